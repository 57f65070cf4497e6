// wp_kernel_common.h — device helpers shared by the kernel translation units that run LDS-resident
// FFTs (wp_conv_kernels.hip: partitioned convolution; wp_longframe_kernels.hip: long-frame convolution).
// Both units are built with -ffp-contract=off -fno-slp-vectorize.
#pragma once
#include <hip/hip_runtime.h>

#include "wp_fft.h"
#include "wp_internal.h"

namespace wp {

// streaming accesses: every spectrum row is written once and read once by a later
// kernel, tens of gigabytes apart, so it is marked non-temporal
typedef float v2f __attribute__((ext_vector_type(2)));
// (WP_PLAIN_SPECTRA builds them as plain accesses: an experiment switch for cache-resident chunked runs)
__device__ __forceinline__ c32 ld_nt(const c32 *p)
{
#if defined(WP_PLAIN_SPECTRA) || defined(WP_PLAIN_SPECTRA_LD)
    return *p;
#else
    const v2f t = __builtin_nontemporal_load(reinterpret_cast<const v2f *>(p));
    return c32{t.x, t.y};
#endif
}
__device__ __forceinline__ void st_nt(c32 *p, c32 v)
{
#if defined(WP_PLAIN_SPECTRA) || defined(WP_PLAIN_SPECTRA_ST)
    *p = v;
#else
    v2f t;
    t.x = v.x;
    t.y = v.y;
    __builtin_nontemporal_store(t, reinterpret_cast<v2f *>(p));
#endif
}

// ---------------------------------------------------------------------------
// Twiddle tables of a kernel.  COPY: staged into LDS by the workgroup (the fused inverse
// kernel, whose register budget has no room for the global loads' latency); otherwise read
// from global memory where they are (one lookup per butterfly and stage, served by L1/L2:
// no staging pass at kernel start, 0.06 ms faster in the forward kernel).
template <int B, bool COPY>
__device__ __forceinline__ TwTables<B> load_tables(const c32 *__restrict__ g, c32 *tws, int tid)
{
    const c32 *base = g;
    if constexpr (COPY) {
        constexpr int NT = B / FftPlan<B>::PT;
        constexpr int N = tw_table_points<B>();
        for (int i = tid; i < N; i += NT) tws[i] = g[i];
        base = tws;
    }
    TwTables<B> tw;
    tw.t256 = base;
    tw.tlo = base + 256;
    tw.thi = base + 512;
    return tw;
}

// All stages but the last store: stage 0 computes from registers, every later stage
// exchanges through LDS; on return v holds the LAST stage's inputs already twiddled and
// transformed (the caller stores them with stage_store<B, LAST, ...>).
template <int B, bool INV>
__device__ __forceinline__ void middle_stages(int tid, c32 *v, c32 *lds, const TwTables<B> &tw)
{
    constexpr int LAST = FftPlan<B>::NSTAGE - 1;
    stage_compute<B, 0, INV>(tid, v, tw);
    static_for<1, LAST + 1>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        stage_store<B, S - 1, INV>(tid, v, [&](int i, c32 val, int, int) { lds[lds_pad(i)] = val; });
        __syncthreads();
        stage_load<B, S>(tid, v, [&](int i) { return lds[lds_pad(i)]; });
        if constexpr (S < LAST) __syncthreads();  // the next store reuses the buffer
        stage_compute<B, S, INV>(tid, v, tw);
    });
}

// middle_stages on the one-pad-per-32 layout (lds_pad32): every LDS address is the butterfly's padded base
// plus a compile-time constant
template <int B, bool INV>
__device__ __forceinline__ void middle_stages32(int tid, c32 *v, c32 *lds, const TwTables<B> &tw)
{
    constexpr int LAST = FftPlan<B>::NSTAGE - 1;
    stage_compute<B, 0, INV>(tid, v, tw);
    static_for<1, LAST + 1>([&](auto sc) {
        constexpr int S = decltype(sc)::value;
        constexpr int NSP = plan_ns<B>(S - 1), STRIDE = B / FftPlan<B>::radix(S);
        stage_store<B, S - 1, INV>(tid, v, [&](int i, c32 val, auto t, auto) {
            lds[lds_pad32(i - int(t) * NSP) + pad32_store_k<NSP>(int(t))] = val;
        });
        __syncthreads();
        stage_load_t<B, S>(tid, v, [&](int j, auto t, auto) { return lds[lds_pad32(j) + pad32_load_k<STRIDE>(int(t))]; });
        if constexpr (S < LAST) __syncthreads();  // the next store reuses the buffer
        stage_compute<B, S, INV>(tid, v, tw);
    });
}

// master-bus fold of one sample (src/mix-buffers.js:49-52): binary64 multiply, binary64 add, binary32 store
__device__ __forceinline__ float mix_fold(float acc, float y, double vol)
{
    return (float)((double)acc + (double)y * vol);  // this file is built with -ffp-contract=off
}
// The same value from one binary32 fused multiply-add when the gain is +-2^k (1, 1/2, 1/256 ...: the usual
// ones).  y*vol is then exact in binary64 with a 24-bit significand, and the binary64 sum of two such numbers
// a, b (|a| >= |b|) is either exact (exponents at most 28 apart), so that the final rounding to binary32 is
// the only one, or b is below 2^-28 |a| and both roundings return a, which is also what the single rounding
// of the exact sum gives.  fmaf rounds the exact acc + y*vol once: the same bits, at a fifth of the
// instructions, all of them binary32 (the binary64 conversions and operations issue at a half to a
// quarter of that rate).
__device__ __forceinline__ bool mix_gain_is_pow2(double vol)
{
    const uint64_t b = (uint64_t)__double_as_longlong(vol);
    const int e = (int)((b >> 52) & 0x7ff);
    return (b & 0x000fffffffffffffULL) == 0 && e >= 1023 - 100 && e <= 1023 + 100;  // exactly representable in binary32 too
}
__device__ __forceinline__ float mix_fold_pow2(float acc, float y, float vol) { return fmaf(y, vol, acc); }


}  // namespace wp
