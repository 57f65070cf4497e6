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

#include "wp_fft.h"
#include "wp_internal.h"

namespace wp {

// ---------------------------------------------------------------------------
template <int B>
__device__ __forceinline__ TwTables<B> load_tables(const c32 *__restrict__ g, c32 *tws, int tid)
{
    constexpr int NT = B / FftPlan<B>::PT;
    constexpr int N = tw_table_points<B>();
    for (int i = tid; i < N; i += NT) tws[i] = g[i];
    TwTables<B> tw;
    tw.t256 = tws;
    tw.tlo = tws + 256;
    tw.thi = tws + 512;
    return tw;
}

// stages 1 and 2 of a three-stage plan, LDS in / LDS exchange between them;
// on return v holds the inputs of stage 2 already twiddled and transformed.
template <int B, bool INV>
__device__ __forceinline__ void middle_stages(int tid, c32 *v, c32 *lds, const TwTables<B> &tw)
{
    stage_compute<B, 0, INV>(tid, v, tw);
    stage_store<B, 0, INV>(tid, v, [&](int i, c32 val, int) { lds[lds_pad(i)] = val; });
    __syncthreads();
    stage_load<B, 1>(tid, v, [&](int i) { return lds[lds_pad(i)]; });
    __syncthreads();
    stage_compute<B, 1, INV>(tid, v, tw);
    stage_store<B, 1, INV>(tid, v, [&](int i, c32 val, int) { lds[lds_pad(i)] = val; });
    __syncthreads();
    stage_load<B, 2>(tid, v, [&](int i) { return lds[lds_pad(i)]; });
    stage_compute<B, 2, INV>(tid, v, tw);
}

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
    const TwTables<B> tw = load_tables<B>(tables, tws, tid);

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
    middle_stages<B, false>(tid, v, lds, tw);
    c32 *z = Z + ((int64_t)job * T + t) * B;
    stage_store<B, 2, false>(tid, v, [&](int i, c32 val, int) { z[i] = val; });
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
    const TwTables<B> tw = load_tables<B>(tables, tws, tid);

    const c32 *w = W + ((int64_t)job * T + t) * B;
    c32 v[FftPlan<B>::PT];
    stage_load<B, 0>(tid, v, [&](int n) { return w[n]; });
    middle_stages<B, true>(tid, v, lds, tw);

    constexpr int R = FftPlan<B>::radix(2);
    float *y = out + (int64_t)job * out_stride;
    const float *pv = prev ? prev + (int64_t)job * prev_stride : nullptr;
    const int64_t o0 = (int64_t)t * B;
    const bool fast = (o0 + B <= out_len) && ((reinterpret_cast<uintptr_t>(y + o0) & 7) == 0) &&
                      (!pv || prev_len <= o0 || (o0 + B <= prev_len && (reinterpret_cast<uintptr_t>(pv + o0) & 7) == 0));
    const bool addp = pv && o0 + B <= prev_len;
    stage_store<B, 2, true>(tid, v, [&](int i, c32 val, auto tt) {
        if constexpr (int(tt) >= R / 2) {  // only the second half of the block is output
            const int64_t o = o0 + 2 * (int64_t)(i - B / 2);
            if (fast) {
                if (addp) {
                    const c32 p = *reinterpret_cast<const c32 *>(pv + o);
                    val.x += p.x;
                    val.y += p.y;
                }
                *reinterpret_cast<c32 *>(y + o) = val;
            } else {
                if (o < out_len) y[o] = val.x + ((pv && o < prev_len) ? pv[o] : 0.0f);
                if (o + 1 < out_len) y[o + 1] = val.y + ((pv && o + 1 < prev_len) ? pv[o + 1] : 0.0f);
            }
        }
    });
}

// ---------------------------------------------------------------------------
// IR partition spectra -> paired layout, scaled by 1/(8B) so that the MAC needs
// no normalisation: Hp[(irch*Pp + p)*(B/2) + q] = (H[q], H[B-q]) for q >= 1 and
// ((H[0], H[B]), H[B/2]) for q == 0, H = R_h / (4B).
__global__ void k_hpack(const c32 *__restrict__ Zir, float4 *__restrict__ Hp, const c32 *__restrict__ wsplit,
                        int B, int P, int Pp)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= B / 2) return;
    const int p = blockIdx.y, ch = blockIdx.z;
    const c32 *z = Zir + ((int64_t)ch * P + p) * B;
    c32 rk, rm;
    if (q == 0) split_q0(z[0], z[B / 2], rk, rm);
    else split_pair(z[q], z[B - q], wsplit[q], rk, rm);
    const float s = 1.0f / (8.0f * (float)B);
    Hp[((int64_t)ch * Pp + p) * (B / 2) + q] = make_float4(rk.x * s, rk.y * s, rm.x * s, rm.y * s);
}

// ---------------------------------------------------------------------------
// Spectral multiply-accumulate across IR partitions (frequency-domain delay
// line held in registers).  Thread q owns bins q and B-q of one job for a run
// of output blocks [ta, tb); it streams Z[tau] once, keeps PG running sums
// (one per in-flight output block) and the PG partition spectra in VGPRs:
//   acc[(i+p) % PG] += split(Z[tau]) * H[p0+p]        for p < PG
// and emits W[t] when the sum for t = tau + p0 is complete.
template <int PG, bool ACCUM>
__global__ __launch_bounds__(256) void k_mac(const c32 *__restrict__ Z, c32 *__restrict__ W,
                                             const float4 *__restrict__ Hp, const c32 *__restrict__ wsplit,
                                             const int32_t *__restrict__ job_irch, int B, int T, int Tin, int Pp,
                                             int p0, int tchunk)
{
    constexpr int D = 4;  // prefetch depth (loads in flight per thread)
    static_assert(PG % D == 0, "ring");
    const int q = blockIdx.x * 256 + threadIdx.x;
    const int half = B >> 1;
    if (q >= half) return;
    const int job = blockIdx.y;
    const int ta = blockIdx.z * tchunk;
    const int tb = (ta + tchunk < T) ? ta + tchunk : T;
    const bool is0 = (q == 0);
    const int km = is0 ? half : B - q;
    const c32 w = wsplit[q];
    const int irch = job_irch[job];

    float4 h[PG];
    {
        const float4 *hp = Hp + ((int64_t)irch * Pp + p0) * half + q;
#pragma unroll
        for (int p = 0; p < PG; p++) h[p] = hp[(int64_t)p * half];
    }
    const c32 *zj = Z + (int64_t)job * T * B;
    c32 *wj = W + (int64_t)job * T * B;

    c32 ak[PG], am[PG];
#pragma unroll
    for (int p = 0; p < PG; p++) ak[p] = am[p] = c32{0.f, 0.f};

    const int tau0 = ta - p0 - (PG - 1);
    const int nsteps = (tb - ta) + (PG - 1);
    auto ld = [&](int i, c32 &a, c32 &b) {
        const int tau = tau0 + i;
        if (i < nsteps && tau >= 0 && tau < Tin) {
            const c32 *z = zj + (int64_t)tau * B;
            a = z[q];
            b = z[km];
        } else {
            a = b = c32{0.f, 0.f};
        }
    };
    c32 rk_[D], rm_[D];
#pragma unroll
    for (int d = 0; d < D; d++) ld(d, rk_[d], rm_[d]);

    for (int i0 = 0; i0 < nsteps; i0 += PG) {
        static_for<0, PG>([&](auto u) {
            const int i = i0 + u;
            if (i < nsteps) {
                constexpr int slot = int(u) % D;
                const c32 zk = rk_[slot], zm = rm_[slot];
                ld(i + D, rk_[slot], rm_[slot]);
                c32 xk, xm;
                if (is0) split_q0(zk, zm, xk, xm);
                else split_pair(zk, zm, w, xk, xm);
                // slot k of thread 0 holds two real bins: elementwise product
                const float xa = is0 ? xk.y : xk.x;
                const float xb = is0 ? 0.0f : xk.y;
                static_for<0, PG>([&](auto p) {
                    constexpr int s = (int(u) + int(p)) % PG;
                    const float4 hh = h[p];
                    ak[s].x += xk.x * hh.x - xb * hh.y;
                    ak[s].y += xa * hh.y + xb * hh.x;
                    am[s].x += xm.x * hh.z - xm.y * hh.w;
                    am[s].y += xm.x * hh.w + xm.y * hh.z;
                });
                const int t = tau0 + i + p0;
                if (i >= PG - 1) {  // t in [ta, tb)
                    c32 ok, om;
                    if (is0) unsplit_q0(ak[u], am[u], ok, om);
                    else unsplit_pair(ak[u], am[u], w, ok, om);
                    c32 *wo = wj + (int64_t)t * B;
                    if (ACCUM) {
                        const c32 a = wo[q], b = wo[km];
                        ok = ok + a;
                        om = om + b;
                    }
                    wo[q] = ok;
                    wo[km] = om;
                }
                ak[u] = am[u] = c32{0.f, 0.f};
            }
        });
    }
}

// ---------------------------------------------------------------------------
// launchers
template <int B>
static hipError_t set_lds_attr(const void *fn)
{
    return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)fft_lds_bytes<B>());
}

template <int B>
static hipError_t launch_fwd_t(bool irmode, const float *in, int64_t in_stride, int64_t in_len, c32 *Z, int T,
                               int nblocks, int njobs, const c32 *tables, hipStream_t s)
{
    if (fft_lds_bytes<B>() > 48 * 1024) {  // opt in to more than the default dynamic LDS
        hipError_t e = irmode ? set_lds_attr<B>((const void *)k_fft_fwd<B, true>)
                              : set_lds_attr<B>((const void *)k_fft_fwd<B, false>);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(nblocks, njobs), block(B / FftPlan<B>::PT);
    if (irmode)
        hipLaunchKernelGGL((k_fft_fwd<B, true>), grid, block, fft_lds_bytes<B>(), s, in, in_stride, in_len, Z, T, tables);
    else
        hipLaunchKernelGGL((k_fft_fwd<B, false>), grid, block, fft_lds_bytes<B>(), s, in, in_stride, in_len, Z, T, tables);
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
    if (fft_lds_bytes<B>() > 48 * 1024) {
        hipError_t e = set_lds_attr<B>((const void *)k_fft_inv<B>);
        if (e != hipSuccess) return e;
    }
    const dim3 grid(T, njobs), block(B / FftPlan<B>::PT);
    hipLaunchKernelGGL((k_fft_inv<B>), grid, block, fft_lds_bytes<B>(), s, W, T, out, out_stride, out_len, prev,
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

hipError_t launch_hpack(const c32 *Zir, float4 *Hp, const c32 *wsplit, int B, int P, int Pp, int nch, hipStream_t s)
{
    const dim3 grid((B / 2 + 255) / 256, P, nch), block(256);
    hipLaunchKernelGGL(k_hpack, grid, block, 0, s, Zir, Hp, wsplit, B, P, Pp);
    return hipGetLastError();
}

template <int PG>
static hipError_t launch_mac_t(const c32 *Z, c32 *W, const float4 *Hp, const c32 *wsplit, const int32_t *job_irch,
                               int B, int T, int Tin, int Pp, int njobs, int tchunk, hipStream_t s)
{
    const int nchunks = (T + tchunk - 1) / tchunk;
    const dim3 grid((B / 2) / 256, njobs, nchunks), block(256);
    for (int p0 = 0; p0 < Pp; p0 += PG) {
        if (p0 == 0)
            hipLaunchKernelGGL((k_mac<PG, false>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, B, T, Tin, Pp, p0, tchunk);
        else
            hipLaunchKernelGGL((k_mac<PG, true>), grid, block, 0, s, Z, W, Hp, wsplit, job_irch, B, T, Tin, Pp, p0, tchunk);
    }
    return hipGetLastError();
}

hipError_t launch_mac(int PG, const c32 *Z, c32 *W, const float4 *Hp, const c32 *wsplit, const int32_t *job_irch,
                      int B, int T, int Tin, int Pp, int njobs, int tchunk, hipStream_t s)
{
    switch (PG) {
    case 4: return launch_mac_t<4>(Z, W, Hp, wsplit, job_irch, B, T, Tin, Pp, njobs, tchunk, s);
    case 8: return launch_mac_t<8>(Z, W, Hp, wsplit, job_irch, B, T, Tin, Pp, njobs, tchunk, s);
    case 12: return launch_mac_t<12>(Z, W, Hp, wsplit, job_irch, B, T, Tin, Pp, njobs, tchunk, s);
    case 16: return launch_mac_t<16>(Z, W, Hp, wsplit, job_irch, B, T, Tin, Pp, njobs, tchunk, s);
    case 24: return launch_mac_t<24>(Z, W, Hp, wsplit, job_irch, B, T, Tin, Pp, njobs, tchunk, s);
    }
    return hipErrorInvalidValue;
}

}  // namespace wp
