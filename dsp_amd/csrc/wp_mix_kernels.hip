// wp_mix_kernels.hip — bandwidth-bound multi-source mix-reduce, the gfx950
// counterpart of src/mix-buffers.js:1-73.
//
// Built with -ffp-contract=off: the reference evaluates
//     target[x] = f32( f64(target[x]) + f64(src[idx]) * vol )
// with a separately rounded binary64 multiply and add per source, in source
// order; doing exactly that here makes the single-GPU result bit-identical.
//
// Layout: one thread owns 4 consecutive target frames (both target channels)
// and walks the sources in order, so every source channel is read exactly once
// with 16-byte accesses and the target is read and written once: the kernel's
// traffic is the algorithmic minimum, 4*(source channels) + 8*(target channels)
// bytes per frame.
#include <hip/hip_runtime.h>
#include <math.h>

#include "wp_internal.h"

namespace wp {

// internal per-source record, produced on the device by k_mix_prep
struct mix_desc {
    const float *ch0;
    const float *ch1;
    const double *vol_vec;
    double vol;
    int64_t rl;
    int64_t asl;   // |sl|
    int64_t o;
    int64_t base;  // contiguous mode: index of frame 0 in the source
    int32_t nch;
    int32_t mode;  // 0 skip, 1 every sample NaN, 2 contiguous, 3 general modular indexing
};
// first record of the scratch area: what k_mix_prep found out about the whole call
struct mix_head {
    int32_t all_fast;  // every source contiguous with a constant gain
    int32_t pad[3];
};

enum { MIX_SKIP = 0, MIX_NAN = 1, MIX_CONTIG = 2, MIX_GENERAL = 3 };

// the device pointers inside a descriptor are global memory: saying so lets the
// compiler emit global_load (vmcnt only) instead of flat_load
typedef const float __attribute__((address_space(1))) *gf32;
typedef const double __attribute__((address_space(1))) *gf64;
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef const f4u __attribute__((address_space(1))) *gf4u;

// src/mix-buffers.js:5-14 decoded once per source; one workgroup walks the table.
__global__ __launch_bounds__(256) void k_mix_prep(const wp_src *__restrict__ srcs, mix_head *__restrict__ head,
                                                  mix_desc *__restrict__ out, int64_t nsrc, int64_t tl, int tch)
{
    int fast = 1;
    for (int64_t s = threadIdx.x; s < nsrc; s += 256) {
        const wp_src S = srcs[s];
        mix_desc d;
        d.ch0 = S.ch[0];
        d.ch1 = S.nch == 2 ? S.ch[1] : S.ch[0];
        d.vol_vec = S.vol_vec;
        d.vol = S.vol;
        d.rl = S.rl;
        d.asl = S.sl < 0 ? -S.sl : S.sl;
        d.o = S.o;
        d.base = 0;
        d.nch = S.nch;
        if ((S.nch != 1 && S.nch != 2) || (tch != 1 && tch != 2)) {
            d.mode = MIX_SKIP;  // the reference silently ignores other layouts (:18-44, :47-69)
        } else if (d.asl == 0 || S.rl <= 0) {
            d.mode = MIX_NAN;   // x % 0 is NaN -> undefined -> NaN
        } else {
            d.mode = MIX_GENERAL;
            if (S.o >= 0) {
                const int64_t om = S.o % d.asl;
                if (om + tl <= d.asl && om + tl <= S.rl) {
                    d.mode = MIX_CONTIG;
                    d.base = om;
                }
            }
        }
        if (d.mode != MIX_CONTIG || d.vol_vec) fast = 0;
        out[s] = d;
    }
    fast = __syncthreads_and(fast);
    if (threadIdx.x == 0) head->all_fast = fast;
}

struct quad {
    float v[4];
};

template <bool FULL>
__device__ __forceinline__ void fetch_contig(gf32 p, int nf, quad &q)
{
    if (FULL) {
        const f4u t = __builtin_nontemporal_load(reinterpret_cast<gf4u>(p));  // streamed once: +10 % over a plain load
        q.v[0] = t.x; q.v[1] = t.y; q.v[2] = t.z; q.v[3] = t.w;
    } else {
#pragma unroll
        for (int f = 0; f < 4; f++) q.v[f] = f < nf ? p[f] : 0.0f;
    }
}

// (x+o) % sl % rl with JavaScript remainder semantics (src/mix-buffers.js:22):
// negative index -> undefined -> NaN, except -0 which names element 0.
__device__ __noinline__ void fetch_general(const mix_desc &d, int64_t x0, int nf, quad &q0, quad &q1)
{
    const float nanv = __builtin_nanf("");
    gf32 c0 = (gf32)d.ch0, c1 = (gf32)d.ch1;
    int64_t v = 0, i = 0;
    bool have = false;
#pragma unroll
    for (int f = 0; f < 4; f++) {
        if (f >= nf) { q0.v[f] = q1.v[f] = 0.0f; continue; }
        const int64_t a = x0 + f + d.o;
        if (a < 0) {
            // a negative remainder is a negative index (NaN) unless it is -0: either a multiple of
            // |sl|, or (|sl| > rl) a remainder that is itself a multiple of rl, e.g. -5 % 10 % 5
            if ((a % d.asl) % d.rl == 0) { q0.v[f] = c0[0]; q1.v[f] = c1[0]; }
            else { q0.v[f] = q1.v[f] = nanv; }
            continue;
        }
        if (!have) {
            v = a % d.asl;
            i = v % d.rl;
            have = true;
        } else {
            v++; i++;
            if (i == d.rl) i = 0;
            if (v == d.asl) { v = 0; i = 0; }
        }
        q0.v[f] = c0[i];
        q1.v[f] = c1[i];
    }
}

// one source folded into the thread's 4 frames (f64 multiply, f64 add, f32 store)
template <int TCH>
__device__ __forceinline__ void accumulate(quad &a0, quad &a1, const quad &q0, const quad &q1, int nch, const double *vol)
{
#pragma unroll
    for (int f = 0; f < 4; f++) {
        if (TCH == 2) {
            if (nch == 2) {  // stereo -> stereo (:22-23, :50-51)
                a0.v[f] = (float)((double)a0.v[f] + (double)q0.v[f] * vol[f]);
                a1.v[f] = (float)((double)a1.v[f] + (double)q1.v[f] * vol[f]);
            } else {         // mono -> stereo (:28-29, :55-56)
                const double h = ((double)q0.v[f] / 2.0) * vol[f];
                a0.v[f] = (float)((double)a0.v[f] + h);
                a1.v[f] = (float)((double)a1.v[f] + h);
            }
        } else {
            if (nch == 2)    // stereo -> mono (:36, :62)
                a0.v[f] = (float)((double)a0.v[f] + (((double)q0.v[f] + (double)q1.v[f]) / 2.0) * vol[f]);
            else             // mono -> mono (:41, :66)
                a0.v[f] = (float)((double)a0.v[f] + (double)q0.v[f] * vol[f]);
        }
    }
}

// FULL: the thread owns 4 whole frames (every thread but possibly the last one)
template <int TCH, bool FULL>
__device__ __forceinline__ void mix_body(float *__restrict__ t0, float *__restrict__ t1, int64_t x0, int nf,
                                         const mix_head *__restrict__ head, const mix_desc *__restrict__ descs,
                                         int64_t nsrc)
{
    quad a0, a1;
    fetch_contig<FULL>((gf32)(t0 + x0), nf, a0);
    if (TCH == 2) fetch_contig<FULL>((gf32)(t1 + x0), nf, a1);

    if (head->all_fast) {
        // every source: contiguous run, constant gain.  U sources' loads are in
        // flight while the (inherently sequential) accumulation drains them.
        constexpr int U = 8;
        int64_t s0 = 0;
        for (; s0 + U <= nsrc; s0 += U) {
            quad q0[U], q1[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const mix_desc &d = descs[s0 + u];
                fetch_contig<FULL>((gf32)(d.ch0 + d.base + x0), nf, q0[u]);
                if (d.nch == 2) fetch_contig<FULL>((gf32)(d.ch1 + d.base + x0), nf, q1[u]);
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const mix_desc &d = descs[s0 + u];
                const double vol[4] = {d.vol, d.vol, d.vol, d.vol};
                accumulate<TCH>(a0, a1, q0[u], q1[u], d.nch, vol);
            }
        }
        for (; s0 < nsrc; s0++) {
            const mix_desc &d = descs[s0];
            quad q0, q1;
            fetch_contig<FULL>((gf32)(d.ch0 + d.base + x0), nf, q0);
            if (d.nch == 2) fetch_contig<FULL>((gf32)(d.ch1 + d.base + x0), nf, q1);
            const double vol[4] = {d.vol, d.vol, d.vol, d.vol};
            accumulate<TCH>(a0, a1, q0, q1, d.nch, vol);
        }
    } else {
        for (int64_t s = 0; s < nsrc; s++) {
            const mix_desc &d = descs[s];
            if (d.mode == MIX_SKIP) continue;
            quad q0, q1;
            if (d.mode == MIX_CONTIG) {
                fetch_contig<FULL>((gf32)(d.ch0 + d.base + x0), nf, q0);
                if (d.nch == 2) fetch_contig<FULL>((gf32)(d.ch1 + d.base + x0), nf, q1);
            } else if (d.mode == MIX_GENERAL) {
                fetch_general(d, x0, nf, q0, q1);
            } else {
#pragma unroll
                for (int f = 0; f < 4; f++) q0.v[f] = q1.v[f] = __builtin_nanf("");
            }
            double vol[4];
            if (d.vol_vec) {
                gf64 vv = (gf64)d.vol_vec;
#pragma unroll
                for (int f = 0; f < 4; f++) vol[f] = f < nf ? vv[x0 + f] : 0.0;
            } else {
#pragma unroll
                for (int f = 0; f < 4; f++) vol[f] = d.vol;
            }
            accumulate<TCH>(a0, a1, q0, q1, d.nch, vol);
        }
    }
    if (FULL) {
        f4u o;
        o.x = a0.v[0]; o.y = a0.v[1]; o.z = a0.v[2]; o.w = a0.v[3];
        *reinterpret_cast<f4u *>(t0 + x0) = o;
        if (TCH == 2) {
            o.x = a1.v[0]; o.y = a1.v[1]; o.z = a1.v[2]; o.w = a1.v[3];
            *reinterpret_cast<f4u *>(t1 + x0) = o;
        }
    } else {
        for (int f = 0; f < nf; f++) {
            t0[x0 + f] = a0.v[f];
            if (TCH == 2) t1[x0 + f] = a1.v[f];
        }
    }
}

// All-contiguous fast path.  A thread owns R rows of 4 frames: the workgroup's tile is
// R * 1024 frames, so every visit to a source reads R coalesced 4 KiB runs of it (with
// non-temporal loads: 6.7 TB/s against 5.9 TB/s for one row and plain loads).
template <int TCH, int R>
__device__ __forceinline__ void mix_rows_fast(float *__restrict__ t0, float *__restrict__ t1, int64_t xb,
                                              const mix_desc *__restrict__ descs, int64_t nsrc)
{
    constexpr int U = 8 / R;  // 16 vector loads in flight per thread either way
    quad a0[R], a1[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        fetch_contig<true>((gf32)(t0 + xb + r * 1024), 4, a0[r]);
        if (TCH == 2) fetch_contig<true>((gf32)(t1 + xb + r * 1024), 4, a1[r]);
    }
    int64_t s0 = 0;
    for (; s0 + U <= nsrc; s0 += U) {
        quad q0[U][R], q1[U][R];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const mix_desc &d = descs[s0 + u];
#pragma unroll
            for (int r = 0; r < R; r++) {
                fetch_contig<true>((gf32)(d.ch0 + d.base + xb + r * 1024), 4, q0[u][r]);
                if (d.nch == 2) fetch_contig<true>((gf32)(d.ch1 + d.base + xb + r * 1024), 4, q1[u][r]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const mix_desc &d = descs[s0 + u];
            const double vol[4] = {d.vol, d.vol, d.vol, d.vol};
#pragma unroll
            for (int r = 0; r < R; r++) accumulate<TCH>(a0[r], a1[r], q0[u][r], q1[u][r], d.nch, vol);
        }
    }
    for (; s0 < nsrc; s0++) {
        const mix_desc &d = descs[s0];
        const double vol[4] = {d.vol, d.vol, d.vol, d.vol};
#pragma unroll
        for (int r = 0; r < R; r++) {
            quad q0, q1;
            fetch_contig<true>((gf32)(d.ch0 + d.base + xb + r * 1024), 4, q0);
            if (d.nch == 2) fetch_contig<true>((gf32)(d.ch1 + d.base + xb + r * 1024), 4, q1);
            accumulate<TCH>(a0[r], a1[r], q0, q1, d.nch, vol);
        }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        f4u o;
        o.x = a0[r].v[0]; o.y = a0[r].v[1]; o.z = a0[r].v[2]; o.w = a0[r].v[3];
        *reinterpret_cast<f4u *>(t0 + xb + r * 1024) = o;
        if (TCH == 2) {
            o.x = a1[r].v[0]; o.y = a1[r].v[1]; o.z = a1[r].v[2]; o.w = a1[r].v[3];
            *reinterpret_cast<f4u *>(t1 + xb + r * 1024) = o;
        }
    }
}

// One workgroup = a tile of R * 1024 frames; thread i owns frames 4i..4i+3 of each
// 1024-frame row of the tile.
template <int TCH, int R>
__global__ __launch_bounds__(256) void k_mix(float *__restrict__ t0, float *__restrict__ t1, int64_t tl,
                                             const mix_head *__restrict__ head,
                                             const mix_desc *__restrict__ descs, int64_t nsrc)
{
    const int64_t tile0 = (int64_t)blockIdx.x * (R * 1024);
    const int64_t xb = tile0 + threadIdx.x * 4;
    if (head->all_fast && tile0 + R * 1024 <= tl) {
        mix_rows_fast<TCH, R>(t0, t1, xb, descs, nsrc);
        return;
    }
    for (int r = 0; r < R; r++) {
        const int64_t x0 = xb + r * 1024;
        if (x0 >= tl) break;
        if (tl - x0 >= 4) mix_body<TCH, true>(t0, t1, x0, 4, head, descs, nsrc);
        else mix_body<TCH, false>(t0, t1, x0, (int)(tl - x0), head, descs, nsrc);
    }
}

size_t mix_scratch_bytes(int64_t nsrc) { return sizeof(mix_head) + sizeof(mix_desc) * (size_t)nsrc; }

hipError_t launch_mix(float *t0, float *t1, int tch, int64_t tl, const wp_src *srcs_dev, void *scratch,
                      int64_t nsrc, hipStream_t s)
{
    if (nsrc <= 0 || tl <= 0) return hipSuccess;
    if (tch != 1 && tch != 2) return hipSuccess;  // the reference does nothing for other target layouts
    mix_head *head = reinterpret_cast<mix_head *>(scratch);
    mix_desc *descs = reinterpret_cast<mix_desc *>(head + 1);
    hipLaunchKernelGGL(k_mix_prep, dim3(1), dim3(256), 0, s, srcs_dev, head, descs, nsrc, tl, tch);
    // long targets: 4-row tiles (fewer, larger visits per source); short ones keep one row per
    // workgroup so that there are enough workgroups
    const bool big = tl >= (int64_t)512 * 4096;
    const int64_t tile = (big ? 4 : 1) * 1024;
    const dim3 grid((unsigned)((tl + tile - 1) / tile)), block(256);
    if (tch == 2) {
        if (big) hipLaunchKernelGGL((k_mix<2, 4>), grid, block, 0, s, t0, t1, tl, head, descs, nsrc);
        else hipLaunchKernelGGL((k_mix<2, 1>), grid, block, 0, s, t0, t1, tl, head, descs, nsrc);
    } else {
        if (big) hipLaunchKernelGGL((k_mix<1, 4>), grid, block, 0, s, t0, t1, tl, head, descs, nsrc);
        else hipLaunchKernelGGL((k_mix<1, 1>), grid, block, 0, s, t0, t1, tl, head, descs, nsrc);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// One source whose samples overlap the target in memory (mixBuffers(t, a, t), a feedback mix
// mixBuffers(t, [t, 1, g, -k]) ...): src/mix-buffers.js:20-67 reads and writes sample by sample, so a read
// may see what earlier iterations of the same loop wrote.  One lane walks the frames in the reference's
// order with every access going to memory: exact for any overlap, and as slow as that sounds (about a
// microsecond per frame); the tiled kernel above takes every source that does not overlap the target.
template <int TCH>
__global__ void k_mix_serial(volatile float *t0, volatile float *t1, int64_t tl, wp_src S)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int64_t asl = S.sl < 0 ? -S.sl : S.sl;
    const volatile float *c0 = S.ch[0], *c1 = S.nch == 2 ? S.ch[1] : S.ch[0];
    const volatile double *vv = S.vol_vec;
    const bool nan_all = asl == 0 || S.rl <= 0;  // x % 0 is NaN -> undefined -> NaN
    const double nanv = (double)__builtin_nanf("");
    for (int64_t x = 0; x < tl; x++) {
        int64_t i = -1;  // -1: the reference reads `undefined` (NaN)
        if (!nan_all) {
            const int64_t a = x + S.o;
            const int64_t r = (a % asl) % S.rl;  // C++ and JavaScript remainders both take the dividend's sign
            i = a >= 0 ? r : (r == 0 ? 0 : -1);  // -0 names element 0
        }
        const double vol = vv ? vv[x] : S.vol;
        if (TCH == 2) {
            if (S.nch == 2) {
                t0[x] = (float)((double)t0[x] + (i < 0 ? nanv : (double)c0[i]) * vol);
                t1[x] = (float)((double)t1[x] + (i < 0 ? nanv : (double)c1[i]) * vol);
            } else {
                t0[x] = (float)((double)t0[x] + ((i < 0 ? nanv : (double)c0[i]) / 2.0) * vol);
                t1[x] = (float)((double)t1[x] + ((i < 0 ? nanv : (double)c0[i]) / 2.0) * vol);
            }
        } else {
            if (S.nch == 2)
                t0[x] = (float)((double)t0[x] + (((i < 0 ? nanv : (double)c0[i]) + (i < 0 ? nanv : (double)c1[i])) / 2.0) * vol);
            else
                t0[x] = (float)((double)t0[x] + (i < 0 ? nanv : (double)c0[i]) * vol);
        }
    }
}

hipError_t launch_mix_serial(float *t0, float *t1, int tch, int64_t tl, const wp_src &src, hipStream_t s)
{
    if (tl <= 0 || (tch != 1 && tch != 2) || (src.nch != 1 && src.nch != 2)) return hipSuccess;
    if (tch == 2)
        hipLaunchKernelGGL((k_mix_serial<2>), dim3(1), dim3(64), 0, s, t0, t1, tl, src);
    else
        hipLaunchKernelGGL((k_mix_serial<1>), dim3(1), dim3(64), 0, s, t0, t1, tl, src);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// synthetic noise: murmur3 finaliser over the sample index, 24-bit mantissa
__global__ void k_synth_noise(float *__restrict__ out, int64_t n, uint32_t k, int64_t start)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (uint32_t)(start + i) ^ k;
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    out[i] = (float)((int32_t)(h >> 8) - 8388608) * (1.0f / 8388608.0f);
}

hipError_t launch_synth_noise(float *out, int64_t n, uint32_t seed, int64_t start, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_noise, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, out, n, seed * 0x9E3779B9u, start);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// circular wrap of a linear convolution result at the reference's FFT length N
__global__ void k_wrap_add(const float *__restrict__ lin, int64_t lin_len, int64_t N, float *__restrict__ out, int64_t out_len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= out_len) return;
    float v = i < lin_len ? lin[i] : 0.0f;
    if (i + N < lin_len) v += lin[i + N];
    out[i] = v;
}

hipError_t launch_wrap_add(const float *lin, int64_t lin_len, int64_t N, float *out, int64_t out_len, hipStream_t s)
{
    if (out_len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_wrap_add, dim3((unsigned)((out_len + 255) / 256)), dim3(256), 0, s, lin, lin_len, N, out, out_len);
    return hipGetLastError();
}

}  // namespace wp
