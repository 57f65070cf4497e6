// wp_conv_kernels.hip — gfx950 kernels for uniformly-partitioned overlap-save
// FFT convolution (the GPU counterpart of lib/convolve.js:553-595 `go`).
//
// Pipeline per channel-job, partition size B, P = ceil(M/B) IR partitions,
// T = ceil((L+M)/B) output blocks:
//   k_fft_fwd   x[(t-1)B,(t+1)B) packed as B complex -> Z[t]      (LDS FFT)
//   k_mac       W[t] = unsplit( sum_p split(Z[t-p]) * H[p] )      (registers)
//   k_fft_inv   W[t] -> last B reals of the inverse -> y[tB,(t+1)B) (+ carried tail)
// All three are HBM-bound streaming kernels; nothing here is a dense
// contraction, so no MFMA.
#include <hip/hip_runtime.h>

#include <atomic>

#include "wp_fft.h"
#include "wp_internal.h"
#include "wp_kernel_common.h"

namespace wp {

// ---------------------------------------------------------------------------
// Forward: one workgroup = one 2B-sample segment of one job.
//   IRMODE = false: segment t covers x[(t-1)B, (t+1)B), samples outside
//                   [0, in_len) read as 0 (overlap-save framing).
//   IRMODE = true : segment p covers h[pB, (p+1)B) followed by B zeros
//                   (IR partition spectra).
template <int B, bool IRMODE>
__global__ __launch_bounds__(B / FftPlan<B>::PT) void k_fft_fwd(const float *__restrict__ in, int64_t in_stride,
                                                               int64_t in_len, c32 *__restrict__ Z, int T,
                                                               const c32 *__restrict__ tables)
{
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    c32 *tws = lds + lds_points<B>();
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    const int job = blockIdx.y;
    const TwTables<B> tw = load_tables<B, false>(tables, tws, tid);

    const float *x = in + (int64_t)job * in_stride;
    const int64_t seg = IRMODE ? (int64_t)t * B : ((int64_t)t - 1) * B;
    const int64_t vlo = IRMODE ? seg : 0;
    const int64_t vhi = IRMODE ? (seg + B < in_len ? seg + B : in_len) : in_len;
    c32 v[FftPlan<B>::PT];
    const bool fast = !IRMODE && seg >= 0 && seg + 2 * B <= in_len && ((reinterpret_cast<uintptr_t>(x + seg) & 7) == 0);
    if (fast) {
        const c32 *xs = reinterpret_cast<const c32 *>(x + seg);
        stage_load<B, 0>(tid, v, [&](int n) { return xs[n]; });
    } else {
        stage_load<B, 0>(tid, v, [&](int n) {
            const int64_t g = seg + 2 * (int64_t)n;
            c32 r;
            r.x = (g >= vlo && g < vhi) ? x[g] : 0.0f;
            r.y = (g + 1 >= vlo && g + 1 < vhi) ? x[g + 1] : 0.0f;
            return r;
        });
    }
    pre_twist<B>(tid, v, tables + table_thalf_offset(B));
    middle_stages<B, false>(tid, v, lds, tw);
    c32 *z = Z + ((int64_t)job * T + t) * B;
    stage_store<B, FftPlan<B>::NSTAGE - 1, false>(tid, v, [&](int i, c32 val, int, int) { st_nt(z + i, val); });
}

// ---------------------------------------------------------------------------
// Forward, two segments per workgroup: segments t0 = 2*blockIdx.x and t0+1 share the B samples
// x[t0*B, (t0+1)*B).  With 16 points per thread and a radix-16 first stage a thread's first eight
// inputs come from the segment's first half and its last eight from the second half, so the
// shared half of segment t0+1 is exactly the thread's own last eight raw inputs of segment t0:
// they stay in registers, and every input sample of a pair is fetched once (three half-segments
// per two transforms instead of four).
template <int B>
__global__ __launch_bounds__(B / FftPlan<B>::PT) void k_fft_fwd_pair(const float *__restrict__ in, int64_t in_stride,
                                                                    int64_t in_len, c32 *__restrict__ Z, int T, int nblocks,
                                                                    const c32 *__restrict__ tables)
{
    constexpr int PT = FftPlan<B>::PT, NT = B / PT, H = PT / 2;
    static_assert(PT == 16 && FftPlan<B>::radix(0) == 16, "a thread's inputs must split into halves of the segment");
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    c32 *tws = lds + lds_points<B>();
    const int tid = threadIdx.x;
    const int t0 = 2 * blockIdx.x;
    const int job = blockIdx.y;
    const bool two = t0 + 1 < nblocks;
    const TwTables<B> tw = load_tables<B, false>(tables, tws, tid);
    const float *x = in + (int64_t)job * in_stride;
    const int64_t seg = ((int64_t)t0 - 1) * B;  // first sample of segment t0; the pair spans [seg, seg + 3B)
    // complex point n of half-segment q (q = 0, 1, 2): samples seg + q*B + 2n, +1
    c32 v[PT], keep[H];
    const int64_t span = two ? 3 * (int64_t)B : 2 * (int64_t)B;
    const bool fast = seg >= 0 && seg + span <= in_len && ((reinterpret_cast<uintptr_t>(x + seg) & 7) == 0);
    auto sample2 = [&](int64_t g) {  // samples g, g+1 with zeros outside [0, in_len)
        c32 r;
        r.x = (g >= 0 && g < in_len) ? x[g] : 0.0f;
        r.y = (g + 1 >= 0 && g + 1 < in_len) ? x[g + 1] : 0.0f;
        return r;
    };
    if (fast) {
        const c32 *xs = reinterpret_cast<const c32 *>(x + seg);
        static_for<0, PT>([&](auto t) { v[t] = xs[tid + int(t) * NT]; });
    } else {
        static_for<0, PT>([&](auto t) { v[t] = sample2(seg + 2 * (int64_t)(tid + int(t) * NT)); });
    }
    static_for<0, H>([&](auto t) { keep[t] = v[H + t]; });
    // the third half-segment is requested now, so that it arrives behind the first transform
    // (fetched only after it, the kernel measured 8 % slower than one segment per workgroup;
    // this way 2.6 % faster)
    c32 next[H];
    if (two) {
        if (fast) {
            const c32 *xs = reinterpret_cast<const c32 *>(x + seg);
            static_for<0, H>([&](auto t) { next[t] = xs[B + tid + int(t) * NT]; });
        } else {
            static_for<0, H>([&](auto t) { next[t] = sample2(seg + 2 * (int64_t)(B + tid + int(t) * NT)); });
        }
    }
    const c32 *thalf = tables + table_thalf_offset(B);
    pre_twist<B>(tid, v, thalf);
    middle_stages<B, false>(tid, v, lds, tw);
    c32 *z = Z + ((int64_t)job * T + t0) * B;
    stage_store<B, FftPlan<B>::NSTAGE - 1, false>(tid, v, [&](int i, c32 val, int, int) { st_nt(z + i, val); });
    if (!two) return;
    static_for<0, H>([&](auto t) { v[H + t] = next[t]; });
    static_for<0, H>([&](auto t) { v[t] = keep[t]; });
    __syncthreads();  // every lane has taken its last-stage inputs out of LDS
    // The twiddles depend on the thread index only.  Left to itself the compiler keeps the first
    // transform's twiddle powers (15 complex values per stage) alive for the second one, which
    // costs some 50 registers; hiding the index behind an empty asm makes it recompute them.
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    pre_twist<B>(tid2, v, thalf);
    middle_stages<B, false>(tid2, v, lds, tw);
    z += B;
    stage_store<B, FftPlan<B>::NSTAGE - 1, false>(tid2, v, [&](int i, c32 val, int, int) { st_nt(z + i, val); });
}

// ---------------------------------------------------------------------------
// Inverse + overlap-save output: W[t] -> y[tB, (t+1)B) = second half of the 2B
// real samples, plus the tail carried from the previous reverb block
// (impulse-reverb.js:14-16).
template <int B>
__global__ __launch_bounds__(B / FftPlan<B>::PT) void k_fft_inv(const c32 *__restrict__ W, int T,
                                                               float *__restrict__ out, int64_t out_stride,
                                                               int64_t out_len, const float *__restrict__ prev,
                                                               int64_t prev_stride, int64_t prev_len,
                                                               const c32 *__restrict__ tables)
{
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    c32 *tws = lds + lds_points<B>();
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    const int job = blockIdx.y;
    const TwTables<B> tw = load_tables<B, false>(tables, tws, tid);

    const c32 *w = W + ((int64_t)job * T + t) * B;
    c32 v[FftPlan<B>::PT];
    stage_load<B, 0>(tid, v, [&](int n) { return ld_nt(w + n); });
    middle_stages<B, true>(tid, v, lds, tw);

    constexpr int LAST = FftPlan<B>::NSTAGE - 1, R = FftPlan<B>::radix(LAST);
    float *y = out + (int64_t)job * out_stride;
    const float *pv = prev ? prev + (int64_t)job * prev_stride : nullptr;
    const int64_t o0 = (int64_t)t * B;
    const bool fast = (o0 + B <= out_len) && ((reinterpret_cast<uintptr_t>(y + o0) & 7) == 0) &&
                      (!pv || prev_len <= o0 || (o0 + B <= prev_len && (reinterpret_cast<uintptr_t>(pv + o0) & 7) == 0));
    const bool addp = pv && o0 + B <= prev_len;
    const c32 *thalf = tables + table_thalf_offset(B);
    stage_store<B, LAST, true>(tid, v, [&](int i, c32 val, auto tt, auto) {
        if constexpr (int(tt) >= R / 2) {  // only the second half of the block is output
            val = post_twist<B, int(tt)>(val, thalf[i & (B / R - 1)]);
            const int64_t o = o0 + 2 * (int64_t)(i - B / 2);
            if (fast) {
                if (addp) {
                    const c32 p = *reinterpret_cast<const c32 *>(pv + o);
                    val.x += p.x;
                    val.y += p.y;
                }
                st_nt(reinterpret_cast<c32 *>(y + o), val);
            } else {
                if (o < out_len) y[o] = val.x + ((pv && o < prev_len) ? pv[o] : 0.0f);
                if (o + 1 < out_len) y[o + 1] = val.y + ((pv && o + 1 < prev_len) ? pv[o + 1] : 0.0f);
            }
        }
    });
}

// ---------------------------------------------------------------------------
// Fused inverse FFT + master-bus mix.  Equivalent to k_fft_inv for every job
// followed by mixBuffers(bus, [[y_0,y_1],1,vol_0,0], [[y_2,y_3],1,vol_1,0], ...)
// (src/mix-buffers.js:49-52, stereo source -> stereo target), without ever
// writing the per-track results: one workgroup owns output block t of one bus channel,
// walks the jobs IN SOURCE ORDER, inverse-transforms W[job][t] and folds the
// samples into register-resident bus accumulators with the reference's
// arithmetic (f64 multiply, f64 add, f32 round per source), so the bus is
// bit-identical to mixing the unfused outputs.  The next job's spectrum is
// loaded while the current one is transformed.
// WRITE: also store every job's samples (the go() results, L+M per job) while folding.
template <int B, bool WRITE>
__global__ __launch_bounds__(B / FftPlan<B>::PT) void k_fft_inv_mix(const c32 *__restrict__ W, int T, int nsrc,
                                                                   float *__restrict__ bus0, float *__restrict__ bus1,
                                                                   int64_t tl, const double *__restrict__ vol,
                                                                   float *__restrict__ out, int64_t out_stride,
                                                                   int64_t out_len, const c32 *__restrict__ tables)
{
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    c32 *tws = lds + lds_points<B>();
    constexpr int LAST = FftPlan<B>::NSTAGE - 1;
    constexpr int PT = FftPlan<B>::PT, R = FftPlan<B>::radix(LAST), NT = B / PT, NS = B / R;
    constexpr int NO = PT / 2;  // complex outputs per thread that belong to the block's second half
    const int tid = threadIdx.x;
    const int t = blockIdx.x;
    const int ch = blockIdx.y;  // the two bus channels are independent sums: one workgroup each
    float *__restrict__ bus = ch ? bus1 : bus0;
    const TwTables<B> tw = load_tables<B, true>(tables, tws, tid);
    const c32 *thalf = tables + table_thalf_offset(B);
    const int64_t o0 = (int64_t)t * B;
    const bool full = (o0 + B <= tl) && ((reinterpret_cast<uintptr_t>(bus + o0) & 7) == 0);
    const bool folds = o0 < tl;  // blocks past the end of the bus only produce per-job samples

    // bus accumulators: sample pair `slot` of this thread sits at o0 + 2*(i - B/2), i = j + tt*NS
    c32 acc[NO];
    auto each_out = [&](auto &&f) {  // f(slot, sample offset) for every kept output of this thread
        static_for<0, PT / R>([&](auto u) {
            static_for<R / 2, R>([&](auto tt) {
                const int i = tid + int(u) * NT + int(tt) * NS;
                f(ic<int(u) * (R / 2) + int(tt) - R / 2>{}, o0 + 2 * (int64_t)(i - B / 2));
            });
        });
    };
    each_out([&](auto slot, int64_t o) {
        if (full) {
            acc[slot] = *reinterpret_cast<const c32 *>(bus + o);
        } else {
            acc[slot].x = o < tl ? bus[o] : 0.f;
            acc[slot].y = o + 1 < tl ? bus[o + 1] : 0.f;
        }
    });

    auto fetch = [&](int src, c32 *v) {  // source index clamped: always a valid row
        const c32 *w = W + ((int64_t)(2 * (src < nsrc ? src : nsrc - 1) + ch) * T + t) * B;
        stage_load<B, 0>(tid, v, [&](int n) { return w[n]; });
    };
    // the post-twist factors of this thread's butterflies (index i mod NS: the same for every output of a
    // butterfly), looked up once: a table load inside the source loop would be younger than the next
    // source's spectrum in flight, and its wait would wait for that too
    c32 th[PT / R];
    static_for<0, PT / R>([&](auto u) { th[u] = thalf[(tid + int(u) * NT) & (NS - 1)]; });
    // FULLW: every source's block of samples lies inside its result and is 8-byte aligned (the same for all
    // sources when the rows are an even number of floats apart): the stores are then unconditional, so the
    // wait for the next source's spectrum can count them instead of waiting for them as well
    auto transform_fold = [&](c32 *v, int src, double g, auto FULLW) {
        middle_stages<B, true>(tid, v, lds, tw);
        float *y = WRITE ? out + (int64_t)(2 * src + ch) * out_stride : nullptr;
        stage_store<B, LAST, true>(tid, v, [&](int i, c32 val, auto tt, auto u) {
            if constexpr (int(tt) >= R / 2) {
                val = post_twist<B, int(tt)>(val, th[u]);
                if constexpr (WRITE) {
                    const int64_t o = o0 + 2 * (int64_t)(i - B / 2);
                    if constexpr (decltype(FULLW)::value) {
                        st_nt(reinterpret_cast<c32 *>(y + o), val);
                    } else {
                        if (o < out_len) y[o] = val.x;
                        if (o + 1 < out_len) y[o + 1] = val.y;
                    }
                }
                if (folds) {
                    constexpr int slot = int(u) * (R / 2) + int(tt) - R / 2;
                    acc[slot].x = mix_fold(acc[slot].x, val.x, g);
                    acc[slot].y = mix_fold(acc[slot].y, val.y, g);
                }
            }
        });
        __syncthreads();  // every lane is done reading LDS before the next transform overwrites it
    };
    const bool fullw_all = WRITE && (o0 + B <= out_len) && (out_stride & 1) == 0 &&
                           (reinterpret_cast<uintptr_t>(out + (int64_t)ch * out_stride + o0) & 7) == 0;

    // two register sets: the next source's spectrum is in flight while this one is transformed
    // (capping the 512-thread plan at 128 VGPRs for a second workgroup per CU measured 2x slower).
    // The twiddle powers depend on the thread index only, so the compiler hoists them out of the
    // source loop and keeps them in some 90 registers (234 in all, two workgroups per CU);
    // recomputing them per source instead (144 registers, three workgroups per CU) measured 3 %
    // slower: the kernel is not bound by occupancy.
    auto run_sources = [&](auto FULLW) {
        if constexpr (PT <= 16) {
            c32 v[PT], vn[PT];
            fetch(0, vn);
#pragma unroll 1
            for (int s = 0; s < nsrc; s++) {
#pragma unroll
                for (int i = 0; i < PT; i++) v[i] = vn[i];
                fetch(s + 1, vn);
                transform_fold(v, s, vol[s], FULLW);
            }
        } else {
            c32 v[PT];
#pragma unroll 1
            for (int s = 0; s < nsrc; s++) {
                fetch(s, v);
                transform_fold(v, s, vol[s], FULLW);
            }
        }
    };
    if (fullw_all)
        run_sources(ic<1>{});
    else
        run_sources(ic<0>{});

    if (folds) each_out([&](auto slot, int64_t o) {
        if (full) {
            *reinterpret_cast<c32 *>(bus + o) = acc[slot];
        } else {
            if (o < tl) bus[o] = acc[slot].x;
            if (o + 1 < tl) bus[o + 1] = acc[slot].y;
        }
    });
}

// ---------------------------------------------------------------------------
// lane layout shared by k_hpack and k_mac: a wavefront owns 32 bin pairs; lanes
// 0-31 hold bins q = 32*wave + lane, lanes 32-63 hold their partners B-1-q, so
// the partner of a lane is lane ^ 32 and each half of a wave accesses one
// 256-byte aligned run.
__device__ __forceinline__ int lane_bin(int slot, int B)
{
    const int q = ((slot >> 6) << 5) + (slot & 31);
    return (slot & 32) ? B - 1 - q : q;
}

// IR partition spectra -> per-bin layout, scaled by 1/(8B) so that the MAC needs
// no normalisation: Hc[(irch*Pp + p)*B + k] = R_h[k]/(4B).
__global__ void k_hpack(const c32 *__restrict__ Zir, c32 *__restrict__ Hc, const c32 *__restrict__ wbin, int B, int P,
                        int Pp)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= B) return;
    const int p = blockIdx.y, ch = blockIdx.z;
    const c32 *z = Zir + ((int64_t)ch * P + p) * B;
    const c32 r = bin_split(z[k], z[B - 1 - k], wbin[k]);
    const float s = 1.0f / (8.0f * (float)B);
    Hc[((int64_t)ch * Pp + p) * B + k] = c32{r.x * s, r.y * s};
}

// loads in flight per lane in k_mac (must divide PG so ring slots stay static)
#ifndef WP_MAC_DEPTH
#define WP_MAC_DEPTH(PG) ((PG) <= 12 ? (PG) : ((PG) == 24 ? 6 : 8))
#endif

// ---------------------------------------------------------------------------
// Spectral multiply-accumulate across IR partitions (frequency-domain delay
// line held in registers).  A lane owns ONE bin of one job for a run of output
// blocks [ta, tb): it streams Z[tau] once, keeps PG running sums (one per
// in-flight output block) and the PG partition spectra of its bin in VGPRs,
//   acc[(i+p) % PG] += split(Z[tau]) * H[p0+p]        for p < PG,
// and emits W[t] when the sum for t = tau + p0 is complete.  The real/packed
// spectrum conversion needs the mirrored bin, which lives in lane ^ 32.
//
// Every load in the steady-state loop is unconditional (out-of-range blocks
// read a clamped address and are zeroed by a select) and every store is
// unconditional, so the compiler can count outstanding memory operations and
// wait with vmcnt(N) instead of draining the queue each iteration.
template <int PG, bool ACCUM>
__global__ __launch_bounds__(256) void k_mac(const c32 *__restrict__ Z, c32 *__restrict__ W,
                                             const c32 *__restrict__ Hc, const c32 *__restrict__ wbin,
                                             const int32_t *__restrict__ job_irch, int irch_all, int B, int T,
                                             int Tout, int Tin, int Pp, int p0, int tchunk)
{
    constexpr int D = (ACCUM && PG == 24) ? 4 : WP_MAC_DEPTH(PG);  // the accumulating pass carries a second ring: keep it at 3 waves per SIMD
    static_assert(PG % D == 0, "ring");
    const int slot = blockIdx.x * 256 + threadIdx.x;  // grid.x * 256 == B
    const int k = lane_bin(slot, B);
    const int job = blockIdx.y;
    const int ta = blockIdx.z * tchunk;
    const int tb = (ta + tchunk < Tout) ? ta + tchunk : Tout;  // only blocks [0, Tout) are needed
    const c32 w = wbin[k];
    const int irch = job_irch ? job_irch[job] : irch_all;  // no table: every job uses channel irch_all

    c32 h[PG];
    {
        const c32 *hp = Hc + ((int64_t)irch * Pp + p0) * B + k;
#pragma unroll
        for (int p = 0; p < PG; p++) h[p] = hp[(int64_t)p * B];
    }
    const c32 *zj = Z + (int64_t)job * T * B + k;
    c32 *wj = W + (int64_t)job * T * B + k;

    c32 acc[PG];
#pragma unroll
    for (int p = 0; p < PG; p++) acc[p] = c32{0.f, 0.f};

    // step i consumes input block tau0 + i and completes output block tau0 + i + p0
    const int tau0 = ta - p0 - (PG - 1);
    const int nsteps = (tb - ta) + (PG - 1);
    const int tin_last = Tin - 1;  // Tin >= 1 (the host launches a memset instead when there is no input)
    auto ld = [&](int i) {         // unconditional: clamped address, zero-selected value
        const int tau = tau0 + i;
        const int tc = tau < 0 ? 0 : (tau > tin_last ? tin_last : tau);
        c32 a = ld_nt(zj + (int64_t)tc * B);
        const bool ok = (tau == tc);
        a.x = ok ? a.x : 0.0f;
        a.y = ok ? a.y : 0.0f;
        return a;
    };
    c32 ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = ld(d);
    // ACCUM (second and later partition groups add into W): the old W rows are
    // prefetched through a ring of their own, D steps ahead, so the
    // read-modify-write never waits on a dependent load.
    const int t_last = tb - 1;
    auto ldw = [&](int i) {
        int tt = tau0 + i + p0;
        tt = tt < ta ? ta : (tt > t_last ? t_last : tt);
        return wj[(int64_t)tt * B];
    };
    c32 wring[D];
    if (ACCUM) {
#pragma unroll
        for (int d = 0; d < D; d++) wring[d] = ldw(d);
    }

    // one step; SLOT = i % PG.  STORE: 0 never (warm-up), 1 always, 2 if i < nsteps
    auto step = [&](int i, auto slot_c, auto store_c) {
        constexpr int u = decltype(slot_c)::value;
        constexpr int STORE = decltype(store_c)::value;
        constexpr int rs = u % D;
        const c32 z = ring[rs];
        ring[rs] = ld(i + D);
        c32 wold = {0.f, 0.f};
        if constexpr (ACCUM) {
            wold = wring[rs];
            wring[rs] = ldw(i + D);
        }
        const c32 zp = {__shfl_xor(z.x, 32), __shfl_xor(z.y, 32)};
        const c32 x = bin_split(z, zp, w);
        static_for<0, PG>([&](auto p) {
            constexpr int s = (u + int(p)) % PG;
            acc[s].x = fmaf(-x.y, h[p].y, fmaf(x.x, h[p].x, acc[s].x));
            acc[s].y = fmaf(x.y, h[p].x, fmaf(x.x, h[p].y, acc[s].y));
        });
        const c32 y = acc[u];
        acc[u] = c32{0.f, 0.f};
        if constexpr (STORE != 0) {
            const c32 yp = {__shfl_xor(y.x, 32), __shfl_xor(y.y, 32)};
            c32 o = bin_unsplit(y, yp, w);
            if (STORE == 1 || i < nsteps) {
                c32 *wo = wj + (int64_t)(tau0 + i + p0) * B;
                if (ACCUM) o = o + wold;
                st_nt(wo, o);
            }
        }
    };

    // warm-up: the first PG-1 steps only fill the delay line
    static_for<0, PG - 1>([&](auto u) { step(int(u), u, ic<0>{}); });
    // steady state: whole blocks of PG steps, everything unconditional
    int i = PG - 1;
    for (; i + PG <= nsteps; i += PG)
        static_for<0, PG>([&](auto j) { step(i + j, ic<(PG - 1 + int(j)) % PG>{}, ic<1>{}); });
    // remainder (fewer than PG steps): stores guarded
    if (i < nsteps)
        static_for<0, PG - 1>([&](auto j) { step(i + j, ic<(PG - 1 + int(j)) % PG>{}, ic<2>{}); });
}

// ---------------------------------------------------------------------------
// The same sum in gather form, for batches too small to fill the chip with the
// streaming kernel above (a stereo reverb block is 2 jobs = 32 workgroups, each a
// serial chain of T + PG steps).  One workgroup = 256 bins of TB consecutive
// output blocks:
//   W[t] = unsplit( sum_p split(Z[t-p]) * H[p] ),   t = t0 .. t0+TB-1,
// partitions taken G at a time: the TB+G-1 block spectra and G partition spectra
// a group needs are independent loads, all in flight together.  Z is re-read
// about Pp/TB times, which is why this form is only chosen while the spectra are
// cache-resident.
template <int TB, int G>
__global__ __launch_bounds__(256) void k_mac_gather(const c32 *__restrict__ Z, c32 *__restrict__ W,
                                                    const c32 *__restrict__ Hc, const c32 *__restrict__ wbin,
                                                    const int32_t *__restrict__ job_irch, int irch_all, int B, int T,
                                                    int Tout, int Tin, int Pp)
{
    constexpr int NZ = TB + G - 1;
    const int slot = blockIdx.x * 256 + threadIdx.x;
    const int k = lane_bin(slot, B);
    const int job = blockIdx.y;
    const int t0 = blockIdx.z * TB;
    const c32 w = wbin[k];
    const c32 *zj = Z + (int64_t)job * T * B + k;
    const c32 *hp = Hc + (int64_t)(job_irch ? job_irch[job] : irch_all) * Pp * B + k;
    const int tin_last = Tin - 1;
    c32 acc[TB];
#pragma unroll
    for (int j = 0; j < TB; j++) acc[j] = c32{0.f, 0.f};
    for (int p0 = 0; p0 < Pp; p0 += G) {  // G divides Pp
        c32 z[NZ], h[G];
#pragma unroll
        for (int e = 0; e < NZ; e++) {  // e = 0 is the newest block this group touches
            const int tau = t0 + (TB - 1) - p0 - e;
            const int tc = tau < 0 ? 0 : (tau > tin_last ? tin_last : tau);
            z[e] = zj[(int64_t)tc * B];
            if (tau != tc) z[e] = c32{0.f, 0.f};
        }
#pragma unroll
        for (int u = 0; u < G; u++) h[u] = hp[(int64_t)(p0 + u) * B];
#pragma unroll
        for (int e = 0; e < NZ; e++) {
            const c32 zp = {__shfl_xor(z[e].x, 32), __shfl_xor(z[e].y, 32)};
            z[e] = bin_split(z[e], zp, w);
        }
#pragma unroll
        for (int j = 0; j < TB; j++) {
#pragma unroll
            for (int u = 0; u < G; u++) {  // block t0+j with partition p0+u reads tau = t0+j-p0-u
                const c32 x = z[(TB - 1) - j + u];
                acc[j].x = fmaf(-x.y, h[u].y, fmaf(x.x, h[u].x, acc[j].x));
                acc[j].y = fmaf(x.y, h[u].x, fmaf(x.x, h[u].y, acc[j].y));
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TB; j++) {
        const c32 yp = {__shfl_xor(acc[j].x, 32), __shfl_xor(acc[j].y, 32)};
        const c32 o = bin_unsplit(acc[j], yp, w);
        if (t0 + j < Tout) W[((int64_t)job * T + t0 + j) * B + k] = o;
    }
}

// ---------------------------------------------------------------------------
// launchers
template <int B>
static hipError_t set_lds_attr(const void *fn)
{
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds_bytes<B>());
}

static std::atomic<bool> g_fwd_pair{true};  // forward transform: two segments per workgroup (false: one)
void set_fwd_pair(bool on) { g_fwd_pair.store(on, std::memory_order_relaxed); }

template <int B>
static hipError_t launch_fwd_t(bool irmode, const float *in, int64_t in_stride, int64_t in_len, c32 *Z, int T,
                               int nblocks, int njobs, const c32 *tables, hipStream_t s)
{
    if ((fft_lds_bytes<B, false>()) > 48 * 1024) {  // opt in to more than the default dynamic LDS
        hipError_t e = irmode ? set_lds_attr<B>((const void *)k_fft_fwd<B, true>)
                              : set_lds_attr<B>((const void *)k_fft_fwd<B, false>);
        if (e == hipSuccess && !irmode) e = set_lds_attr<B>((const void *)k_fft_fwd_pair<B>);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(nblocks, njobs), block(B / FftPlan<B>::PT);
    if (irmode)
        hipLaunchKernelGGL((k_fft_fwd<B, true>), grid, block, (fft_lds_bytes<B, false>()), s, in, in_stride, in_len, Z, T, tables);
    else if (g_fwd_pair.load(std::memory_order_relaxed) && nblocks > 1 && (int64_t)nblocks * njobs >= 4096)  // (a launch that does not fill the chip is over sooner with a workgroup per segment; same bits)
        hipLaunchKernelGGL((k_fft_fwd_pair<B>), dim3((nblocks + 1) / 2, njobs), block, (fft_lds_bytes<B, false>()), s, in,
                           in_stride, in_len, Z, T, nblocks, tables);
    else
        hipLaunchKernelGGL((k_fft_fwd<B, false>), grid, block, (fft_lds_bytes<B, false>()), s, in, in_stride, in_len, Z, T, tables);
    return hipGetLastError();
}

hipError_t launch_fft_fwd(int B, bool irmode, const float *in, int64_t in_stride, int64_t in_len, c32 *Z, int T,
                          int nblocks, int njobs, const c32 *tables, hipStream_t s)
{
    switch (B) {
    case 1024: return launch_fwd_t<1024>(irmode, in, in_stride, in_len, Z, T, nblocks, njobs, tables, s);
    case 2048: return launch_fwd_t<2048>(irmode, in, in_stride, in_len, Z, T, nblocks, njobs, tables, s);
    case 4096: return launch_fwd_t<4096>(irmode, in, in_stride, in_len, Z, T, nblocks, njobs, tables, s);
    case 8192: return launch_fwd_t<8192>(irmode, in, in_stride, in_len, Z, T, nblocks, njobs, tables, s);
    }
    return hipErrorInvalidValue;
}

template <int B>
static hipError_t launch_inv_t(const c32 *W, int T, int njobs, float *out, int64_t out_stride, int64_t out_len,
                               const float *prev, int64_t prev_stride, int64_t prev_len, const c32 *tables,
                               hipStream_t s)
{
    if ((fft_lds_bytes<B, false>()) > 48 * 1024) {
        hipError_t e = set_lds_attr<B>((const void *)k_fft_inv<B>);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(T, njobs), block(B / FftPlan<B>::PT);
    hipLaunchKernelGGL((k_fft_inv<B>), grid, block, (fft_lds_bytes<B, false>()), s, W, T, out, out_stride, out_len, prev,
                       prev_stride, prev_len, tables);
    return hipGetLastError();
}

hipError_t launch_fft_inv(int B, const c32 *W, int T, int njobs, float *out, int64_t out_stride, int64_t out_len,
                          const float *prev, int64_t prev_stride, int64_t prev_len, const c32 *tables, hipStream_t s)
{
    switch (B) {
    case 1024: return launch_inv_t<1024>(W, T, njobs, out, out_stride, out_len, prev, prev_stride, prev_len, tables, s);
    case 2048: return launch_inv_t<2048>(W, T, njobs, out, out_stride, out_len, prev, prev_stride, prev_len, tables, s);
    case 4096: return launch_inv_t<4096>(W, T, njobs, out, out_stride, out_len, prev, prev_stride, prev_len, tables, s);
    case 8192: return launch_inv_t<8192>(W, T, njobs, out, out_stride, out_len, prev, prev_stride, prev_len, tables, s);
    }
    return hipErrorInvalidValue;
}

template <int B>
static hipError_t launch_inv_mix_t(const c32 *W, int T, int nblocks, int nsrc, float *bus0, float *bus1, int64_t tl,
                                   const double *vol, float *out, int64_t out_stride, int64_t out_len,
                                   const c32 *tables, hipStream_t s)
{
    if (fft_lds_bytes<B>() > 48 * 1024) {
        hipError_t e = out ? set_lds_attr<B>((const void *)k_fft_inv_mix<B, true>)
                           : set_lds_attr<B>((const void *)k_fft_inv_mix<B, false>);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(nblocks, 2), block(B / FftPlan<B>::PT);
    if (out)
        hipLaunchKernelGGL((k_fft_inv_mix<B, true>), grid, block, fft_lds_bytes<B>(), s, W, T, nsrc, bus0, bus1, tl, vol,
                           out, out_stride, out_len, tables);
    else
        hipLaunchKernelGGL((k_fft_inv_mix<B, false>), grid, block, fft_lds_bytes<B>(), s, W, T, nsrc, bus0, bus1, tl, vol,
                           out, out_stride, out_len, tables);
    return hipGetLastError();
}

hipError_t launch_fft_inv_mix(int B, const c32 *W, int T, int nblocks, int nsrc, float *bus0, float *bus1, int64_t tl,
                              const double *vol, float *out, int64_t out_stride, int64_t out_len, const c32 *tables,
                              hipStream_t s)
{
    switch (B) {
    case 1024: return launch_inv_mix_t<1024>(W, T, nblocks, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tables, s);
    case 2048: return launch_inv_mix_t<2048>(W, T, nblocks, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tables, s);
    case 4096: return launch_inv_mix_t<4096>(W, T, nblocks, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tables, s);
    case 8192: return launch_inv_mix_t<8192>(W, T, nblocks, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tables, s);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_hpack(const c32 *Zir, c32 *Hc, const c32 *wbin, int B, int P, int Pp, int nch, hipStream_t s)
{
    const dim3 grid(B / 256, P, nch), block(256);
    hipLaunchKernelGGL(k_hpack, grid, block, 0, s, Zir, Hc, wbin, B, P, Pp);
    return hipGetLastError();
}

template <int PG>
static hipError_t launch_mac_t(const c32 *Z, c32 *W, const c32 *Hp, const c32 *wsplit, const int32_t *job_irch,
                               int irch_all, int B, int T, int Tout, int Tin, int Pp, int njobs, int tchunk, hipStream_t s)
{
    const int nchunks = (Tout + tchunk - 1) / tchunk;
    const dim3 grid(B / 256, njobs, nchunks), block(256);
    for (int p0 = 0; p0 < Pp; p0 += PG) {
        if (p0 == 0)
            hipLaunchKernelGGL((k_mac<PG, false>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, p0, tchunk);
        else
            hipLaunchKernelGGL((k_mac<PG, true>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, p0, tchunk);
    }
    return hipGetLastError();
}

hipError_t launch_mac(int PG, const c32 *Z, c32 *W, const c32 *Hp, const c32 *wsplit, const int32_t *job_irch,
                      int irch_all, int B, int T, int Tout, int Tin, int Pp, int njobs, int tchunk, hipStream_t s)
{
    switch (PG) {
    case 4: return launch_mac_t<4>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, tchunk, s);
    case 8: return launch_mac_t<8>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, tchunk, s);
    case 12: return launch_mac_t<12>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, tchunk, s);
    case 16: return launch_mac_t<16>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, tchunk, s);
    case 24: return launch_mac_t<24>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, tchunk, s);
    }
    return hipErrorInvalidValue;
}

template <int TB>
static hipError_t launch_mac_gather_t(const c32 *Z, c32 *W, const c32 *Hp, const c32 *wsplit, const int32_t *job_irch,
                                      int irch_all, int B, int T, int Tout, int Tin, int Pp, int njobs, hipStream_t s)
{
    const dim3 grid(B / 256, njobs, (Tout + TB - 1) / TB), block(256);
    if (Pp % 24 == 0)
        hipLaunchKernelGGL((k_mac_gather<TB, 24>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp);
    else if (Pp % 12 == 0)
        hipLaunchKernelGGL((k_mac_gather<TB, 12>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp);
    else if (Pp % 8 == 0)
        hipLaunchKernelGGL((k_mac_gather<TB, 8>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp);
    else
        hipLaunchKernelGGL((k_mac_gather<TB, 4>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp);
    return hipGetLastError();
}

hipError_t launch_mac_gather(const c32 *Z, c32 *W, const c32 *Hp, const c32 *wsplit, const int32_t *job_irch,
                             int irch_all, int B, int T, int Tout, int Tin, int Pp, int njobs, hipStream_t s)
{
    if (Tout <= 0) return hipSuccess;
    if (Tout > 65535 || njobs > 65535 || Pp % 4) return hipErrorInvalidValue;
    // more output blocks per workgroup share more loads; keep at least a workgroup per CU
    const int64_t wgs1 = (int64_t)(B / 256) * njobs * Tout;
    if (wgs1 >= 4 * 256) return launch_mac_gather_t<4>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, s);
    if (wgs1 >= 2 * 256) return launch_mac_gather_t<2>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, s);
    return launch_mac_gather_t<1>(Z, W, Hp, wsplit, job_irch, irch_all, B, T, Tout, Tin, Pp, njobs, s);
}

}  // namespace wp
