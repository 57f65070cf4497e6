// wp_longframe_kernels.hip — long-frame overlap-save convolution for long inputs
// (the GPU counterpart of one `go` of lib/convolve.js:564-586 when L is many times M).
//
// The partitioned form (wp_conv_kernels.hip) writes and reads a block spectrum of 8 bytes per
// sample twice (Z and W): 40 bytes per output sample.  For inputs much longer than the impulse
// response the same linear convolution is cheaper as overlap-save with frames of N2 = 2*Nc real
// samples that each hold the WHOLE impulse response: a frame advances by hop = N2 - Mpad samples
// (Mpad >= M-1) and costs 20 bytes per frame sample plus 4 per output sample (26 bytes per output
// sample for a 60 s track and a 2 s response covered by three frames of 2^20 samples and one of
// 2^18), and there is no multiply-accumulate over partitions.  A result is covered by up to three
// runs of equal frames (wp_api.hip: choose_lf); every kernel here handles one run per launch.
//
// A frame is packed as Nc complex points z[m] = x[2m] + i x[2m+1], pre-twisted by exp(-i*pi*m/Nc)
// (odd-frequency transform, wp_fft.h) and transformed by a four-step FFT, Nc = R * C, m = r*C + c:
//   k_lf_fwd   R-point transforms down the columns (tiles of 32 columns x R rows, 256-byte runs),
//              times exp(-2*pi*i*c*k1/Nc)                              -> A[k1][c]
//   k_lf_mid, k_lf_mid_jobs
//              per row pair (k1, R-1-k1): C-point transform along the row gives bins k1 + R*k2;
//              the partner bin Nc-1-k sits in the other row at C-1-k2, so the real/packed split,
//              the product with the kernel spectrum, the unsplit and the inverse C-point
//              transform all happen in one workgroup, in place (k_lf_mid_jobs: for eight jobs in
//              turn, the next job's rows in flight)
//   k_lf_inv   conj twiddle, inverse R-point transforms down the columns, post-twist, the
//              frame's last hop samples written out; with MIX the workgroup walks the sources
//              in order and folds the samples into the master bus (src/mix-buffers.js:49-52)
//              exactly as k_fft_inv_mix does.
// A[k1][c] is stored in blocks of 32 rows x chunks of 64 columns (wp_internal.h: lf_row_off, lf_col_off).
// All three are HBM-bound streaming kernels; nothing here is a dense contraction (no MFMA).
#include <hip/hip_runtime.h>

#include <atomic>

#include "wp_fft.h"
#include "wp_internal.h"
#include "wp_kernel_common.h"

namespace wp {

// (tiles of the column passes are COLS columns wide: 32 complex points = one 256-byte run per row)

// exp(-2*pi*i*idx/Nc), idx < Nc, from the two-level table (both factors correctly rounded)
__device__ __forceinline__ c32 lf_w(const c32 *__restrict__ tab, int idx)
{
    return cmul(tab[LF_FHI + (idx >> 9)], tab[LF_FLO + (idx & 511)]);
}

// p[t] = w^t, t < N, by a product tree (each power at most log2(t) products from the lookup)
template <int N>
__device__ __forceinline__ void lf_powers(c32 w, c32 *p)
{
    p[0] = c32{1.0f, 0.0f};
    p[1] = w;
    static_for<2, N>([&](auto t) { p[t] = cmul(p[int(t) / 2], p[int(t) - int(t) / 2]); });
}

template <int R>
__device__ __forceinline__ TwTables<R> lf_col_tables(const c32 *__restrict__ tab)
{
    TwTables<R> tw;  // every stage of a transform of at most 256 points looks up t256 only
    tw.t256 = tab + LF_T256;
    tw.tlo = tab + LF_T256;
    tw.thi = tab + LF_T256;
    return tw;
}

// ---------------------------------------------------------------------------
// Forward column pass.  Workgroup = tile of 32 columns of one frame of one job; thread = (column,
// j) with R/16 threads per column transform.  IRMODE: the "frame" is the impulse response itself,
// starting at sample 0.
template <int R, int C, int COLS>
__global__ __launch_bounds__(COLS *R / 16) void k_lf_fwd(const float *__restrict__ in, int64_t in_stride, int64_t in_len,
                                                            c32 *__restrict__ A, int nframes, int64_t base, int64_t hop,
                                                            int64_t mpad, const c32 *__restrict__ tab)
{
    constexpr int NTF = R / 16, PT = 16;
    constexpr int64_t Nc = (int64_t)R * C;
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    const int col = threadIdx.x & (COLS - 1), j = threadIdx.x / COLS;
    const int c = blockIdx.x * COLS + col;
    const int frame = blockIdx.y, job = blockIdx.z;
    const TwTables<R> tw = lf_col_tables<R>(tab);
    const float *x = in + (int64_t)job * in_stride;
    const int64_t start = base + (int64_t)frame * hop - mpad;  // sample at frame position 0 (base: first output sample of this run of frames)
    // When the frame starts on an 8-byte boundary (the same for every pair of it), a sample pair is one 8-byte load
    // through a buffer descriptor over the job's input: the hardware returns zeros for pairs outside it (offsets of
    // pairs ahead of the input wrap to out of range), so a frame that starts ahead of the input or ends behind it
    // (the first and the last of a result) loads like any other.  A pair that straddles the input's end is patched.
    const bool al = (reinterpret_cast<uintptr_t>(x + start) & 7) == 0 && in_len < ((int64_t)1 << 29);  // byte offsets of every pair, in or out of range, stay below 2^32
    c32 v[PT];
    if (al) {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(x), 0, (uint32_t)in_len * 4u, 0x00020000);
        static_for<0, PT>([&](auto t) {
            const int64_t g = start + 2 * ((int64_t)(j + int(t) * NTF) * C + c);
            const v2f p = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(xrs, (int)((uint32_t)g * 4u), 0, 0));
            v[t] = c32{p.x, p.y};
        });
        static_for<0, PT>([&](auto t) {
            const int64_t g = start + 2 * ((int64_t)(j + int(t) * NTF) * C + c);
            if (g + 1 == in_len) v[t] = c32{x[g], 0.0f};
        });
    } else {
        static_for<0, PT>([&](auto t) {
            const int64_t g = start + 2 * ((int64_t)(j + int(t) * NTF) * C + c);
            v[t].x = (g >= 0 && g < in_len) ? x[g] : 0.0f;
            v[t].y = (g + 1 >= 0 && g + 1 < in_len) ? x[g + 1] : 0.0f;
        });
    }
    // pre-twist exp(-i*pi*m/Nc), m = r*C + c, r = j + t*NTF:  exp(-i*pi*j/R) * exp(-i*pi*c/Nc) * exp(-i*pi*t/16)
    {
        const c32 f = cmul(tab[LF_ROW + j], tab[LF_COL + c]);
        static_for<0, PT>([&](auto t) { v[t] = cmul(v[t], mul_w32<int(t), false>(f)); });
    }
#if !defined(WP_FWD_ABLATE)  // (measurement builds: the samples go out as they came in, through the same loads and stores)
    stage_compute<R, 0, false>(j, v, tw);
    stage_store<R, 0, false>(j, v, [&](int i, c32 val, int, int) { lds[i * COLS + col] = val; });
    __syncthreads();
    stage_load<R, 1>(j, v, [&](int i) { return lds[i * COLS + col]; });
    stage_compute<R, 1, false>(j, v, tw);
#endif
    // outputs k1 = j' + 16*t (j' = j + u*NTF), times exp(-2*pi*i*c*k1/Nc) = W^(c*j') * (W^(16c))^t
    constexpr int R1 = FftPlan<R>::radix(1), NB = PT / R1;
    c32 pw[R1], wj[NB];
    lf_powers<R1>(lf_w(tab, 16 * c), pw);
    static_for<0, NB>([&](auto u) { wj[u] = lf_w(tab, c * (j + int(u) * NTF)); });
    c32 *a = A + ((int64_t)job * nframes + frame) * Nc + lf_col_off(c);
    stage_store<R, 1, false>(j, v, [&](int k1, c32 val, auto t, auto u) {
        // k1 = j' + 16*t with j' < 16: the row offset is a constant of t plus 64*j'
        st_nt(a + (k1 - 16 * int(t)) * LF_CC + lf_row_off(16 * int(t), C), cmul(val, cmul(wj[u], pw[t])));
    });
}

// ---------------------------------------------------------------------------
// Row pass.  Workgroup = rows k1 and R-1-k1 of one frame of one job (256 threads: two C-point
// transforms of 128 threads each).  HPREP: the rows are an impulse response's; store its real
// spectrum scaled by 1/(8*Nc) (the factors 2 of split and unsplit and the inverse transform's Nc)
// and stop.
//
// AB (kernel spectra shared by the jobs, i.e. read from cache): the chain split -> product -> unsplit of a
// bin pair is linear in (za, conj zb), and with w = exp(-i*theta) its coefficients collapse to
//   Ya = alpha_a*za + beta_a*conj(zb),   alpha_a = 2(1 - sin)*Ha + 2(1 + sin)*conj(Hb),  beta_a = 2i*cos*(Ha - conj(Hb))
//   Yb = alpha_b*zb + beta_b*conj(za),   alpha_b = 2(1 - sin)*Hb + 2(1 + sin)*conj(Ha),  beta_b = -conj(beta_a)
// so the spectra are stored as (alpha, beta) per bin (16 bytes) and a pair costs 16 instructions instead of 44.
// Kernel spectra that every job reads for itself from HBM keep the 8-byte form.
struct alignas(16) c32x2 {
    c32 a, b;
};
// alpha*z + beta*conj(zp)
__device__ __forceinline__ c32 lf_pair_product(c32x2 h, c32 z, c32 zp)
{
    c32 y;
    y.x = fmaf(h.b.y, zp.y, fmaf(h.b.x, zp.x, fmaf(h.a.x, z.x, -(h.a.y * z.y))));
    y.y = fmaf(-h.b.x, zp.y, fmaf(h.b.y, zp.x, fmaf(h.a.x, z.y, h.a.y * z.x)));
    return y;
}

template <bool HPREP, bool AB, int C>
__global__ __launch_bounds__(C / 8) void k_lf_mid(c32 *__restrict__ A, c32 *__restrict__ Hs, const int32_t *__restrict__ job_irch,
                                                int irch_all, int R, int nframes, const c32 *__restrict__ tab,
                                                const c32 *__restrict__ tab_row)
{
    constexpr int NT = C / 16, WGT = 2 * NT;  // threads per row transform, per workgroup
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    c32 *tws = lds + 2 * C;  // not filled: this kernel's row transforms read their tables from global memory
    const int tid = threadIdx.x, half = tid / NT, ft = tid & (NT - 1);
    const int job = blockIdx.x, k1 = blockIdx.y, frame = blockIdx.z;  // jobs fastest: neighbours share the kernel spectrum rows
    const int64_t Nc = (int64_t)R * C;
    const int row = half ? R - 1 - k1 : k1;
    c32 *rowp = A + ((int64_t)job * nframes + frame) * R * C + lf_row_off(row, C);  // point n of the row: rowp + lf_col_off(n)
    c32 *ldsrow = lds + half * C;
    const TwTables<C> tw = load_tables<C, false>(tab_row, tws, ft);

    c32 v[16];
    stage_load<C, 0>(ft, v, [&](int n) { return ld_nt(rowp + lf_col_off(n)); });
    const int irch = HPREP ? job : (job_irch ? job_irch[job] : irch_all);
    // plain spectra lie like a frame of A (a job that reads its own from HBM streams them the same way)
    c32 *ha = AB ? Hs + ((int64_t)irch * R + k1) * C * 2 : Hs + (int64_t)irch * R * C + lf_row_off(k1, C);
    c32 *hb = AB ? Hs + ((int64_t)irch * R + (R - 1 - k1)) * C * 2 : Hs + (int64_t)irch * R * C + lf_row_off(R - 1 - k1, C);
    middle_stages<C, false>(ft, v, ldsrow, tw);
    __syncthreads();  // every lane has taken its last-stage inputs out of LDS
    stage_store<C, FftPlan<C>::NSTAGE - 1, false>(ft, v, [&](int i, c32 val, int, int) { ldsrow[lds_pad(i)] = val; });
    __syncthreads();

    // bins k = k1 + R*k2 (row k1) and their partners Nc-1-k = (R-1-k1) + R*(C-1-k2) (the other row)
    const c32 wa = tab[LF_WA + k1];
    const float scale = 1.0f / (8.0f * (float)Nc);
#pragma unroll
    for (int q = 0; q < C / WGT; q++) {
        const int k2 = tid + WGT * q, p2 = C - 1 - k2;
        const c32 za = lds[lds_pad(k2)], zb = lds[C + lds_pad(p2)];
        if constexpr (!HPREP && AB) {
            const c32x2 pa = reinterpret_cast<const c32x2 *>(ha)[k2], pb = reinterpret_cast<const c32x2 *>(hb)[p2];
            lds[lds_pad(k2)] = lf_pair_product(pa, za, zb);
            lds[C + lds_pad(p2)] = lf_pair_product(pb, zb, za);
        } else {
            const c32 wk = cmul(wa, tab[LF_WB + k2]);  // exp(-i*pi*(k+1/2)/Nc)
            const c32 wp = c32{-wk.x, wk.y};          // the partner's: -conj(wk)
            const c32 xa = bin_split(za, zb, wk), xb = bin_split(zb, za, wp);
            if constexpr (HPREP) {
                const c32 a = c32{xa.x * scale, xa.y * scale}, b = c32{xb.x * scale, xb.y * scale};
                if constexpr (AB) {
                    const float sn = -wk.y, cs = wk.x, m = 2.0f * (1.0f - sn), p = 2.0f * (1.0f + sn);
                    c32x2 pa, pb;
                    pa.a = c32{fmaf(m, a.x, p * b.x), fmaf(m, a.y, -(p * b.y))};
                    pb.a = c32{fmaf(m, b.x, p * a.x), fmaf(m, b.y, -(p * a.y))};
                    pa.b = c32{-2.0f * cs * (a.y + b.y), 2.0f * cs * (a.x - b.x)};  // 2i*cos*(Ha - conj(Hb))
                    pb.b = c32{-pa.b.x, pa.b.y};                                    // -conj(beta_a)
                    reinterpret_cast<c32x2 *>(ha)[k2] = pa;
                    reinterpret_cast<c32x2 *>(hb)[p2] = pb;
                } else {
                    ha[lf_col_off(k2)] = a;
                    hb[lf_col_off(p2)] = b;
                }
            } else {
                const c32 ya = cmul(xa, ha[lf_col_off(k2)]), yb = cmul(xb, hb[lf_col_off(p2)]);
                lds[lds_pad(k2)] = bin_unsplit(ya, yb, wk);
                lds[C + lds_pad(p2)] = bin_unsplit(yb, ya, wp);
            }
        }
    }
    if (HPREP) return;
    __syncthreads();
    stage_load<C, 0>(ft, v, [&](int n) { return ldsrow[lds_pad(n)]; });
    __syncthreads();  // all stage-0 inputs are in registers before the first exchange overwrites the row
    int fi = ft;  // hidden from the optimiser, which would otherwise keep the forward transform's twiddle powers alive for the inverse (119 instead of 76 registers)
    asm volatile("" : "+v"(fi));
    middle_stages<C, true>(fi, v, ldsrow, tw);
    stage_store<C, FftPlan<C>::NSTAGE - 1, true>(fi, v, [&](int i, c32 val, int, int) { st_nt(rowp + lf_col_off(i), val); });
}

// The row pass for batches: a workgroup walks JB consecutive jobs of its row pair with the next job's
// rows already on their way while the current ones are transformed, so the latency of the 32 KB it
// streams per job is covered by its own work instead of by other workgroups.  Inside the loop nothing
// else is loaded from global memory ahead of the kernel spectrum (a younger load's wait would also wait
// for the rows in flight): the row plan's tables are copied to LDS, the split factors kept in registers.
// Same arithmetic as k_lf_mid<false>.
template <int JB, bool AB, int C>
__global__ __launch_bounds__(C / 8) void k_lf_mid_jobs(c32 *__restrict__ A, const c32 *__restrict__ Hs,
                                                     const int32_t *__restrict__ job_irch, int irch_all, int R, int nframes,
                                                     int njobs, const c32 *__restrict__ tab, const c32 *__restrict__ tab_row)
{
    constexpr int NT = C / 16, WGT = 2 * NT, LAST = FftPlan<C>::NSTAGE - 1, NSL = plan_ns<C>(LAST);
    static_assert(FftPlan<C>::radix(0) == 16 && FftPlan<C>::PT == 16, "one stage-0 butterfly per thread");
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    constexpr int CP = lds_points32<C>();  // a row in LDS: one pad element per 32 (lds_pad32)
    c32 *tws = lds + 2 * CP;
    const int tid = threadIdx.x, half = tid / NT;
    const int ft = pad32_lane_butterfly(tid & (NT - 1));  // the butterfly this lane takes: keeps every LDS exchange free of bank conflicts
    const int k1 = blockIdx.y, frame = blockIdx.z;
    const int job0 = blockIdx.x * JB, jend = job0 + JB < njobs ? job0 + JB : njobs;
    const int64_t Nc = (int64_t)R * C;
    const int row = half ? R - 1 - k1 : k1;
    c32 *ldsrow = lds + half * CP;
    const TwTables<C> tw = load_tables<C, true>(tab_row, tws, tid);  // first used behind the first exchange's barrier
    // a row access = a base that is the same in every lane (job, frame, point group; kept in scalar registers by
    // an empty asm, or the optimiser folds it into sixteen per-lane 64-bit offsets) + one per-lane offset
    c32 *__restrict__ Arow = A + lf_row_off(row, C);
    auto fetch = [&](int job, c32 *v) {
        static_for<0, 16>([&](auto t) {  // point ft + NT*t: chunk (ft >> 6) + (NT/64)*t, column ft & 63 of it
            int64_t at = ((int64_t)job * nframes + frame) * Nc + lf_col_off(int(t) * (C / 16));
            asm volatile("" : "+s"(at));
            v[t] = ld_nt(Arow + lf_col_off(ft) + at);
        });
    };
    // The same rows through a buffer descriptor (the shared-spectra form): an unconditional load whose descriptor
    // is empty behind the last job returns zeros without touching memory, so the registers of the rows in flight
    // are defined anew in every pass and free between the top of a pass and the request.
    const int row_u = __builtin_amdgcn_readfirstlane(row);  // the same in every lane of a wave (NT is a multiple of 64)
    const c32 *__restrict__ Arow_u = A + lf_row_off(row_u, C);
    constexpr uint32_t ROW_SPAN = (uint32_t)(lf_col_off(C - 1) + 1) * 8u;  // bytes from a row's first point to the end of its last
    const int voff = lf_col_off(ft) * 8;
    auto fetch_desc = [&](int job, bool live, c32 *v) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<c32 *>(Arow_u + ((int64_t)job * nframes + frame) * Nc), 0, live ? ROW_SPAN : 0u, 0x00020000);
        static_for<0, 16>([&](auto t) {
            const v2f p = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, lf_col_off(int(t) * (C / 16)) * 8, 2 /* non-temporal */));
            v[t] = c32{p.x, p.y};
        });
    };
    const c32 wa = tab[LF_WA + k1];
    c32 wk[AB ? 1 : C / WGT];  // exp(-i*pi*(k+1/2)/Nc) of this thread's bins k = k1 + R*k2, k2 = tid + WGT*q
    if constexpr (!AB) static_for<0, C / WGT>([&](auto q) { wk[q] = cmul(wa, tab[LF_WB + tid + WGT * int(q)]); });

    c32 v[16], vn[16];
    fetch(job0, vn);
    // The first job's rows are waited for here, ahead of the loop: with loads still in flight on entry and
    // only stores on the way round, the wait the compiler places at the loop's head would have to cover
    // both, i.e. drain every pass's stores before the next pass starts.
#pragma unroll
    for (int t = 0; t < 16; t++) asm volatile("" ::"v"(vn[t].x), "v"(vn[t].y));
#pragma unroll 1
    for (int job = job0; job < jend; job++) {
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = vn[i];
        const int irch = job_irch ? job_irch[job] : irch_all;
        // the thread indices are hidden from the optimiser once per job (and once more between the two
        // transforms): every LDS address and twiddle power derived from them would otherwise be computed
        // once, ahead of the loop, and held in some 150 registers across it
        int ff = ft, tl = tid;
        asm volatile("" : "+v"(ff), "+v"(tl));
        // A kernel spectrum of the job's own comes from HBM: its bins k2 = tl + WGT*q of row k1 and C-1-k2 of the
        // other row are requested here, a whole forward transform ahead of their use (they lie like a frame of A).
        c32 hA[AB ? 1 : C / WGT], hB[AB ? 1 : C / WGT];
        if constexpr (!AB) {
            const c32 *__restrict__ ha = Hs + (int64_t)irch * R * C + lf_row_off(k1, C) + lf_col_off(tl);
            const c32 *__restrict__ hb = Hs + (int64_t)irch * R * C + lf_row_off(R - 1 - k1, C) + lf_col_off(C - 1 - tl);
            static_for<0, C / WGT>([&](auto q) {
                hA[q] = ld_nt(ha + lf_col_off(WGT * int(q)));
                hB[q] = ld_nt(hb - lf_col_off(WGT * int(q)));
            });
        }
#if defined(WP_MID_ABLATE)  // measurement builds only: 1 = rows copied through registers, 2 = forward transform only
        if (WP_MID_ABLATE == 2) middle_stages32<C, false>(ff, v, ldsrow, tw);
        stage_store<C, LAST, false>(ff, v, [&](int i, c32 val, auto t, auto) {
#if defined(WP_MID_ABLATE_SHIFT)  // results written to another job's rows: the stream is no longer in place (timing only)
            int64_t at = ((int64_t)((job + njobs / 2) % njobs) * nframes + frame) * Nc + lf_col_off(int(t) * NSL);
#else
            int64_t at = ((int64_t)job * nframes + frame) * Nc + lf_col_off(int(t) * NSL);
#endif
            asm volatile("" : "+s"(at));
            st_nt(Arow + lf_col_off(i - int(t) * NSL) + at, val);
        });
        __syncthreads();
        continue;
#endif
        middle_stages32<C, false>(ff, v, ldsrow, tw);
        int fi = ft;  // the inverse transform's copy of the index, hidden like ff
        asm volatile("" : "+v"(fi));
        if constexpr (AB) {
            // Own-bin form.  The forward transform leaves this thread with elements ff + NT*m of its row, m < 16:
            // exactly the inputs of its inverse stage-0 butterfly.  Only their partners C-1-(ff + NT*m) of the
            // other row have to come through LDS, and Y = alpha*z + beta*conj(z_partner) needs this bin's
            // coefficients alone: 16 LDS stores and 16 loads per thread instead of 32 and 32, one barrier fewer.
            // All sixteen coefficient loads are issued before the barriers (they are served by L2: their latency
            // passes behind the exchange) and AHEAD of the next job's rows, so that waiting for them never waits
            // for those rows; the rows are requested when half the coefficients have been used up.
            const c32x2 *__restrict__ hrow = reinterpret_cast<const c32x2 *>(Hs + ((int64_t)irch * R + row) * C * 2) + ff;
            c32x2 co[16];
            static_for<0, 16>([&](auto m) { co[m] = hrow[NT * int(m)]; });
            __syncthreads();  // every lane has taken its last-stage inputs out of LDS
            stage_store<C, LAST, false>(ff, v, [&](int i, c32 val, auto t, auto) {
                ldsrow[lds_pad32(i - int(t) * NSL) + pad32_store_k<NSL>(int(t))] = val;
            });
            __syncthreads();
            constexpr int RL = FftPlan<C>::radix(LAST), NBL = 16 / RL;
            c32 w[16];  // w[m] = element ff + NT*m (stage_store's output t of butterfly u is element ff + NT*(u + NBL*t))
            static_for<0, NBL>([&](auto u) {
                static_for<0, RL>([&](auto t) { w[int(u) + NBL * int(t)] = v[int(u) * RL + Dft<RL, false>::perm(int(t))]; });
            });
            const c32 *zb_p = lds + (half ? 0 : CP) + lds_pad32(C - 1 - ff);  // partner of element ff + NT*m: zb_p[-pad32_load_k<NT>(m)]
            constexpr int FETCH_AT = 8;  // products done before the next job's rows are requested (4, 8 and 12 measure the same)
            static_for<0, FETCH_AT>([&](auto m) { v[m] = lf_pair_product(co[m], w[m], zb_p[-pad32_load_k<NT>(int(m))]); });
            asm volatile("" ::: "memory");
            fetch_desc(job + 1, job + 1 < jend, vn);
            asm volatile("" ::: "memory");
            static_for<FETCH_AT, 16>([&](auto m) { v[m] = lf_pair_product(co[m], w[m], zb_p[-pad32_load_k<NT>(int(m))]); });
            __syncthreads();  // every partner has been read before the inverse transform's first exchange overwrites the rows
        } else {
            __syncthreads();  // every lane has taken its last-stage inputs out of LDS
            stage_store<C, LAST, false>(ff, v, [&](int i, c32 val, auto t, auto) {
                ldsrow[lds_pad32(i - int(t) * NSL) + pad32_store_k<NSL>(int(t))] = val;
            });
            __syncthreads();
            // bins k2 = tl + WGT*q of row k1 and their partners C-1-k2 of the other row: WGT + WGT/32 padded positions per q
            c32 *za_p = lds + lds_pad32(tl), *zb_p = lds + CP + lds_pad32(C - 1 - tl);
            auto pairs = [&](auto q) {
                c32 &za_s = za_p[(WGT + WGT / 32) * int(q)], &zb_s = zb_p[-(WGT + WGT / 32) * int(q)];
                const c32 za = za_s, zb = zb_s;
                const c32 wp = c32{-wk[q].x, wk[q].y};  // the partner's factor: -conj(wk)
                const c32 xa = bin_split(za, zb, wk[q]), xb = bin_split(zb, za, wp);
                const c32 ya = cmul(xa, hA[q]), yb = cmul(xb, hB[q]);
                za_s = bin_unsplit(ya, yb, wk[q]);
                zb_s = bin_unsplit(yb, ya, wp);
            };
            static_for<0, C / WGT / 2>(pairs);
            asm volatile("" ::: "memory");
            fetch_desc(job + 1, job + 1 < jend, vn);  // behind this job's kernel-spectrum loads: waiting for those never waits for these
            asm volatile("" ::: "memory");
            static_for<C / WGT / 2, C / WGT>(pairs);
            __syncthreads();
            stage_load_t<C, 0>(fi, v, [&](int j, auto t, auto) { return ldsrow[lds_pad32(j) + pad32_load_k<C / 16>(int(t))]; });
            __syncthreads();  // all stage-0 inputs are in registers before the first exchange overwrites the row
        }
        middle_stages32<C, true>(fi, v, ldsrow, tw);
        stage_store<C, LAST, true>(fi, v, [&](int i, c32 val, auto t, auto) {
            int64_t at = ((int64_t)job * nframes + frame) * Nc + lf_col_off(int(t) * NSL);
            asm volatile("" : "+s"(at));
            st_nt(Arow + lf_col_off(i - int(t) * NSL) + at, val);
        });
        __syncthreads();  // the next job's first exchange writes the rows this one's last exchange read
    }
}

// ---------------------------------------------------------------------------
// Inverse column pass + output (+ master-bus fold).  MIX: workgroup = (tile, frame) of ONE bus
// channel (blockIdx.z), walking the stereo sources in order (jobs 2s + channel) with the next
// source's tile prefetched; otherwise blockIdx.z is the job.  WRITE: store the job's samples.
// COLS: columns per tile (32, or 16 for R = 256 so that the workgroup stays at 256 threads: with 512 the
// bus-fold variants need more registers than two waves per SIMD leave and spill).
template <int R, int C, int COLS, bool MIX, bool WRITE>
__global__ __launch_bounds__(COLS *R / 16) void k_lf_inv(const c32 *__restrict__ A, int nframes, int frame0, int64_t base, int64_t hop,
                                                            int64_t mpad, int nsrc, float *__restrict__ bus0, float *__restrict__ bus1,
                                                            int64_t tl, const double *__restrict__ vol, float *__restrict__ out,
                                                            int64_t out_stride, int64_t out_len, const c32 *__restrict__ tab)
{
    constexpr int NTF = R / 16, PT = 16;
    constexpr int64_t Nc = (int64_t)R * C;
    constexpr int R1 = FftPlan<R>::radix(1), NB = PT / R1;
    extern __shared__ __align__(16) unsigned char smem[];
    c32 *lds = reinterpret_cast<c32 *>(smem);
    const int col = threadIdx.x & (COLS - 1), j = threadIdx.x / COLS;
    const int c = blockIdx.x * COLS + col;
    const int frame = frame0 + blockIdx.y;
    const TwTables<R> tw = lf_col_tables<R>(tab);
    // inputs k1 = j + t*NTF: conj of exp(-2*pi*i*c*k1/Nc) = conj(W^(c*j)) * conj(W^(c*NTF))^t
    c32 tin[PT];
    {
        c32 pw[PT];
        lf_powers<PT>(cconj(lf_w(tab, c * NTF)), pw);
        const c32 w0 = cconj(lf_w(tab, c * j));
        static_for<0, PT>([&](auto t) { tin[t] = cmul(w0, pw[t]); });
    }
    // outputs r = j' + 16*t (j' = j + u*NTF): post-twist exp(+i*pi*m/Nc), m = r*C + c
    c32 pf[NB];
    static_for<0, NB>([&](auto u) { pf[u] = cconj(cmul(tab[LF_ROW + j + int(u) * NTF], tab[LF_COL + c])); });
    // sample pair of output (u, t): frame position 2m, kept when it lies in the frame's last hop samples
    const int64_t obase = base + (int64_t)frame * hop - mpad + 2 * (int64_t)c;
    auto out_index = [&](int r) { return obase + 2 * (int64_t)r * C; };  // global sample index (valid when >= frame*hop)
    const int64_t first = base + (int64_t)frame * hop;

    // A tile access = a base that is the same in every lane (job, frame, row group: scalar registers, passed
    // through an empty asm so that the optimiser does not fold it into sixteen per-lane 64-bit offsets that
    // would live across the source loop) + one per-lane offset.
    static_assert(LF_RB % NTF == 0, "rows j + t*NTF, j < NTF: the row offset splits into a constant of t and LF_CC*j");
    const c32 *__restrict__ At = A + (j * LF_CC + lf_col_off(c));
    auto fetch = [&](int job, c32 *v) {
        static_for<0, PT>([&](auto t) {
            int64_t at = ((int64_t)job * nframes + frame) * Nc + lf_row_off(int(t) * NTF, C);
            asm volatile("" : "+s"(at));
            v[t] = ld_nt(At + at);
        });
    };
    // the second stage's looked-up twiddles, taken out of the source loop (a table load inside it would
    // make every pass wait for the next source's tile, which is in flight behind it)
    c32 w1[NB];
    stage_first_powers<R, 1, true>(j, w1, tw);
    auto transform = [&](c32 *v) {
        static_for<0, PT>([&](auto t) { v[t] = cmul(v[t], tin[t]); });
        stage_compute<R, 0, true>(j, v, tw);
        stage_store<R, 0, true>(j, v, [&](int i, c32 val, int, int) { lds[i * COLS + col] = val; });
        __syncthreads();
        stage_load<R, 1>(j, v, [&](int i) { return lds[i * COLS + col]; });
        stage_compute_with<R, 1, true>(v, w1);
    };

    if constexpr (!MIX) {
        const int job = blockIdx.z;
        c32 v[PT];
        fetch(job, v);
        transform(v);
        float *y = out + (int64_t)job * out_stride;
        const bool al = (reinterpret_cast<uintptr_t>(y) & 7) == 0;  // the sample index is even
        stage_store<R, 1, true>(j, v, [&](int r, c32 val, auto t, auto u) {
            val = cmul(val, mul_w32<int(t) * (256 / R), true>(pf[u]));
            const int64_t o = out_index(r);
            if (o >= first) {
                if (al && o + 1 < out_len) {
                    st_nt(reinterpret_cast<c32 *>(y + o), val);
                } else {
                    if (o < out_len) y[o] = val.x;
                    if (o + 1 < out_len) y[o + 1] = val.y;
                }
            }
        });
    } else {
        const int ch = blockIdx.z;
        float *__restrict__ bus = ch ? bus1 : bus0;
        const bool bal = (reinterpret_cast<uintptr_t>(bus) & 7) == 0;
        c32 acc[PT];
        auto each_out = [&](auto &&f) {  // f(slot, global sample index) for every output of this thread
            static_for<0, NB>([&](auto u) {
                static_for<0, R1>([&](auto t) {
                    f(ic<int(u) * R1 + int(t)>{}, out_index(j + int(u) * NTF + int(t) * 16));
                });
            });
        };
        each_out([&](auto slot, int64_t o) {
            acc[slot] = c32{0.f, 0.f};
            if (o >= first) {
                if (bal && o + 1 < tl) {
                    acc[slot] = *reinterpret_cast<const c32 *>(bus + o);
                } else {
                    if (o < tl) acc[slot].x = bus[o];
                    if (o + 1 < tl) acc[slot].y = bus[o + 1];
                }
            }
        });
        // A source's samples go out through a buffer descriptor over the part of its result that this
        // frame owns, [first, first + room): the store is one unconditional instruction per pair (the
        // hardware drops lanes whose byte offset, negative ones wrapped, lies outside the descriptor), so
        // the wait for the next source's tile counts the stores issued behind its loads instead of
        // waiting for them too.  A pair that straddles the result's end (odd length) takes the `edge` path.
        int64_t room = out_len - first;
        if (room > hop) room = hop;
        if (room < 0 || !WRITE) room = 0;
        const uint32_t nrec = (uint32_t)(room & ~(int64_t)1) * 4u;
        int ybo[NB];  // byte offset of output row j + u*NTF (t = 0) from `first`
        static_for<0, NB>([&](auto u) { ybo[u] = (2 * (c + (j + int(u) * NTF) * C) - (int)mpad) * 4; });
        // bit slot of `live`: the pair reaches the result or the bus; of `edge`: only its first sample is inside the result
        uint32_t live = 0, edge = 0;
        const int64_t lim = WRITE && out_len > tl ? out_len : tl;
        each_out([&](auto slot, int64_t o) {
            if (o >= first) {
                if (o < lim) live |= 1u << int(slot);
                if (WRITE && o + 1 == out_len) edge |= 1u << int(slot);
            }
        });
        // pairs of a slot that is live in some lane of the wave are folded in every lane: accumulators of
        // pairs outside the frame or the bus are never stored
        uint32_t wl = 0;
        static_for<0, PT>([&](auto slot) {
            if (__builtin_amdgcn_ballot_w64((live >> int(slot)) & 1u) != 0) wl |= 1u << int(slot);
        });
        const uint32_t wlive = __builtin_amdgcn_readfirstlane(wl);
        const bool any_edge = __builtin_amdgcn_ballot_w64(edge != 0) != 0;
        // the input twiddles wait in LDS behind the exchange tile, one private slot per thread and point:
        // 32 registers fewer across the source loop
        constexpr int NT = COLS * R / 16;
        c32 *tinL = lds + R * COLS + threadIdx.x;
        static_for<0, PT>([&](auto t) { tinL[int(t) * NT] = tin[t]; });
        c32 v[PT], vn[PT];
        fetch(ch, vn);
#pragma unroll 1
        for (int s = 0; s < nsrc; s++) {
#pragma unroll
            for (int i = 0; i < PT; i++) v[i] = vn[i];
            fetch(2 * (s + 1 < nsrc ? s + 1 : nsrc - 1) + ch, vn);  // clamped: always a valid tile
            const double g = vol[s];
#if !defined(WP_INV_ABLATE)  // measurement builds: the tile goes out as it came in (same loads, stores and fold)
            static_for<0, PT>([&](auto t) { v[t] = cmul(v[t], tinL[int(t) * NT]); });
            stage_compute<R, 0, true>(j, v, tw);
            stage_store<R, 0, true>(j, v, [&](int i, c32 val, int, int) { lds[i * COLS + col] = val; });
            __syncthreads();
            stage_load<R, 1>(j, v, [&](int i) { return lds[i * COLS + col]; });
#endif
#if !defined(WP_INV_ABLATE)
            {
                c32 wl[NB];  // hidden from the optimiser: the powers derived from them would otherwise live across the loop
                static_for<0, NB>([&](auto u) {
                    wl[u] = w1[u];
                    asm volatile("" : "+v"(wl[u].x), "+v"(wl[u].y));
                });
                stage_compute_with<R, 1, true>(v, wl);
            }
#endif
            float *yb = WRITE ? out + (int64_t)(2 * s + ch) * out_stride + first : nullptr;
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(yb, 0, nrec, 0x00020000);
            const float g32 = (float)g;
            auto finish = [&](auto POW2) {  // post-twist, the source's samples out, the fold into the bus
                stage_store<R, 1, true>(j, v, [&](int, c32 val, auto t, auto u) {
                    constexpr int slot = int(u) * R1 + int(t);
                    if (wlive & (1u << slot)) {
                        val = cmul(val, mul_w32<int(t) * (256 / R), true>(pf[u]));
                        if constexpr (WRITE) {
                            const int off = ybo[u] + 2 * C * 16 * 4 * int(t);
                            __builtin_amdgcn_raw_buffer_store_b64(v2f{val.x, val.y}, yrs, off, 0, 2 /* non-temporal */);
                            if (any_edge && (edge & (1u << slot))) yb[off >> 2] = val.x;
                        }
                        if constexpr (decltype(POW2)::value) {
                            acc[slot].x = mix_fold_pow2(acc[slot].x, val.x, g32);
                            acc[slot].y = mix_fold_pow2(acc[slot].y, val.y, g32);
                        } else {
                            acc[slot].x = mix_fold(acc[slot].x, val.x, g);
                            acc[slot].y = mix_fold(acc[slot].y, val.y, g);
                        }
                    }
                });
            };
            if (mix_gain_is_pow2(g))  // the same in every lane
                finish(ic<1>{});
            else
                finish(ic<0>{});
            __syncthreads();  // every lane is done reading LDS before the next transform overwrites it
        }
        each_out([&](auto slot, int64_t o) {
            if (o >= first) {
                if (bal && o + 1 < tl) {
                    *reinterpret_cast<c32 *>(bus + o) = acc[slot];
                } else {
                    if (o < tl) bus[o] = acc[slot].x;
                    if (o + 1 < tl) bus[o + 1] = acc[slot].y;
                }
            }
        });
    }
}

// ---------------------------------------------------------------------------
// launchers
// (wp_tune may change it while other threads launch: a relaxed atomic, read once per launch)
static std::atomic<int> g_mid_jobs{8};  // batches of at least 16 jobs: the row pass that walks 8 jobs per workgroup (0: one job)
void set_mid_jobs(int jobs) { g_mid_jobs.store(jobs, std::memory_order_relaxed); }
template <class K>
static hipError_t lf_lds(K kernel, size_t bytes)
{
    if (bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

// frame shapes (rows x columns) the kernels are instantiated for: 2^17 ... 2^21 samples
bool lf_shape_ok(int R, int C) { return (C == LF_C && (R == 32 || R == 64 || R == 128 || R == 256)) || (C == LF_C_BIG && R == 256); }
// f(ic<R>, ic<C>) for the shape given at run time
template <class F>
static hipError_t lf_dispatch(int R, int C, F &&f)
{
    if (C == LF_C_BIG && R == 256) return f(ic<256>{}, ic<LF_C_BIG>{});
    if (C != LF_C) return hipErrorInvalidValue;
    switch (R) {
    case 32: return f(ic<32>{}, ic<LF_C>{});
    case 64: return f(ic<64>{}, ic<LF_C>{});
    case 128: return f(ic<128>{}, ic<LF_C>{});
    case 256: return f(ic<256>{}, ic<LF_C>{});
    }
    return hipErrorInvalidValue;
}

template <int R, int C>
static hipError_t lf_fwd_t(const float *in, int64_t in_stride, int64_t in_len, c32 *A, int nframes, int frames_run, int njobs,
                           int64_t base, int64_t hop, int64_t mpad, const c32 *tab, hipStream_t s)
{
#ifndef WP_FWD_COLS
#define WP_FWD_COLS 32
#endif
    constexpr int FC = WP_FWD_COLS * R / 16 > 1024 ? 1024 * 16 / R : WP_FWD_COLS;  // columns per tile
    const size_t lds = sizeof(c32) * R * FC;
    if (hipError_t e = lf_lds(k_lf_fwd<R, C, FC>, lds)) return e;
    hipLaunchKernelGGL((k_lf_fwd<R, C, FC>), dim3(C / FC, frames_run, njobs), dim3(FC * R / 16), lds, s, in, in_stride, in_len,
                       A, nframes, base, hop, mpad, tab);
    return hipGetLastError();
}

hipError_t launch_lf_fwd(int R, int C, const float *in, int64_t in_stride, int64_t in_len, c32 *A, int nframes, int frames_run,
                         int njobs, int64_t base, int64_t hop, int64_t mpad, const c32 *tab, hipStream_t s)
{
    if (frames_run <= 0) return hipSuccess;
    if (nframes > 65535 || njobs > 65535) return hipErrorInvalidValue;
    return lf_dispatch(R, C, [&](auto r, auto c) {
        return lf_fwd_t<int(r), int(c)>(in, in_stride, in_len, A, nframes, frames_run, njobs, base, hop, mpad, tab, s);
    });
}

template <int C>
static hipError_t lf_mid_t(bool hprep, bool ab, int R, c32 *A, c32 *Hs, const int32_t *job_irch, int irch_all, int nframes,
                           int frames_run, int njobs, const c32 *tab, const c32 *tab_row, hipStream_t s)
{
    const dim3 block(C / 8);
    if (!hprep && njobs >= 16 && g_mid_jobs.load(std::memory_order_relaxed)) {
        constexpr int JB = 8;  // jobs a workgroup walks (measured with the own-bin form: 4 -> +1.4 %, 16 -> +12 %)
        const size_t ldsj = sizeof(c32) * (2 * lds_points32<C>() + tw_table_points<C>());
        const dim3 grid((njobs + JB - 1) / JB, R / 2, frames_run);
        if (ab) {
            if (hipError_t e = lf_lds(k_lf_mid_jobs<JB, true, C>, ldsj)) return e;
            hipLaunchKernelGGL((k_lf_mid_jobs<JB, true, C>), grid, block, ldsj, s, A, Hs, job_irch, irch_all, R, nframes, njobs, tab, tab_row);
        } else {
            if (hipError_t e = lf_lds(k_lf_mid_jobs<JB, false, C>, ldsj)) return e;
            hipLaunchKernelGGL((k_lf_mid_jobs<JB, false, C>), grid, block, ldsj, s, A, Hs, job_irch, irch_all, R, nframes, njobs, tab, tab_row);
        }
        return hipGetLastError();
    }
    const size_t lds = sizeof(c32) * 2 * C;
    const dim3 grid(njobs, R / 2, frames_run);
#define WP_MID(H, B)                                                                                                              \
    do {                                                                                                                          \
        if (hipError_t e = lf_lds(k_lf_mid<H, B, C>, lds)) return e;                                                              \
        hipLaunchKernelGGL((k_lf_mid<H, B, C>), grid, block, lds, s, A, Hs, job_irch, irch_all, R, nframes, tab, tab_row);        \
    } while (0)
    if (hprep && ab)
        WP_MID(true, true);
    else if (hprep)
        WP_MID(true, false);
    else if (ab)
        WP_MID(false, true);
    else
        WP_MID(false, false);
#undef WP_MID
    return hipGetLastError();
}

hipError_t launch_lf_mid(bool hprep, bool ab, int R, int C, c32 *A, c32 *Hs, const int32_t *job_irch, int irch_all, int nframes,
                         int frames_run, int njobs, const c32 *tab, const c32 *tab_row, hipStream_t s)
{
    if (frames_run <= 0) return hipSuccess;
    if (!lf_shape_ok(R, C)) return hipErrorInvalidValue;
    if (C == LF_C_BIG) return lf_mid_t<LF_C_BIG>(hprep, ab, R, A, Hs, job_irch, irch_all, nframes, frames_run, njobs, tab, tab_row, s);
    return lf_mid_t<LF_C>(hprep, ab, R, A, Hs, job_irch, irch_all, nframes, frames_run, njobs, tab, tab_row, s);
}

// compute units of a device (asked once per device)
static int lf_cu_count()
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

// Columns per tile of the bus-fold variants.  Frames of 256 rows: 16 (workgroups of 256 threads, 128-byte runs)
// or, `wide`, 32 (512 threads, 256-byte runs: a chain then takes 1.5 times as long for twice the samples, but only
// one workgroup fits a CU).  Smaller frames: 32.
template <int R, int C, int COLS>
static hipError_t lf_inv_fold_t(const c32 *A, int nframes, int frame0, int frames_run, int64_t base, int64_t hop, int64_t mpad, int nsrc,
                                float *bus0, float *bus1, int64_t tl, const double *vol, float *out, int64_t out_stride, int64_t out_len,
                                const c32 *tab, hipStream_t s)
{
    size_t lds = sizeof(c32) * R * COLS * 2;  // the bus-fold variants keep their input twiddles behind the tile
    const dim3 block(COLS * R / 16);
    if (out && block.x == 256) {
        // (measured for the variant that also writes the sources' samples; the bus-only one gains from two per CU)
        // A bus-fold workgroup is one long serial chain over the sources, and two of them on a CU run at half
        // speed each: the pass takes (rounds of workgroups) x (workgroups per CU).  Two per CU fit; when one per
        // CU needs fewer such units (e.g. 768 workgroups on 256 CUs: 3 against 2 x 2), ask for more LDS than
        // two workgroups can share.
        const int ncu = lf_cu_count();
        const int64_t units = (int64_t)(C / COLS) * frames_run * 2;
        const int64_t one = (units + ncu - 1) / ncu, two = 2 * ((units + 2 * ncu - 1) / (2 * ncu));
        if (one < two && lds < 84 * 1024) lds = 84 * 1024;
    }
    if (out) {
        if (hipError_t e = lf_lds(k_lf_inv<R, C, COLS, true, true>, lds)) return e;
        hipLaunchKernelGGL((k_lf_inv<R, C, COLS, true, true>), dim3(C / COLS, frames_run, 2), block, lds, s, A, nframes, frame0, base, hop, mpad,
                           nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tab);
    } else {
        if (hipError_t e = lf_lds(k_lf_inv<R, C, COLS, true, false>, lds)) return e;
        hipLaunchKernelGGL((k_lf_inv<R, C, COLS, true, false>), dim3(C / COLS, frames_run, 2), block, lds, s, A, nframes, frame0, base, hop, mpad,
                           nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tab);
    }
    return hipGetLastError();
}

template <int R, int C>
static hipError_t lf_inv_t(const c32 *A, int nframes, int frame0, int frames_run, int njobs, int64_t base, int64_t hop, int64_t mpad, int nsrc,
                           float *bus0, float *bus1, int64_t tl, const double *vol, float *out, int64_t out_stride, int64_t out_len,
                           const c32 *tab, bool wide, hipStream_t s)
{
#ifndef WP_INVP_COLS
#define WP_INVP_COLS 32
#endif
    if (!bus0) {  // the plain pass has registers to spare for larger workgroups: tiles of 32 columns (256-byte runs) for every frame size
        constexpr int PC = WP_INVP_COLS * R / 16 > 1024 ? 1024 * 16 / R : WP_INVP_COLS;
        const size_t ldsp = sizeof(c32) * R * PC;
        if (hipError_t e = lf_lds(k_lf_inv<R, C, PC, false, true>, ldsp)) return e;
        hipLaunchKernelGGL((k_lf_inv<R, C, PC, false, true>), dim3(C / PC, frames_run, njobs), dim3(PC * R / 16), ldsp, s, A, nframes, frame0, base,
                           hop, mpad, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tab);
        return hipGetLastError();
    }
    if constexpr (R > 128) {
        if (wide)
            return lf_inv_fold_t<R, C, 32>(A, nframes, frame0, frames_run, base, hop, mpad, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tab, s);
        return lf_inv_fold_t<R, C, 16>(A, nframes, frame0, frames_run, base, hop, mpad, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tab, s);
    } else {
        return lf_inv_fold_t<R, C, 32>(A, nframes, frame0, frames_run, base, hop, mpad, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len, tab, s);
    }
}

// bus0 == nullptr: plain results for every job; otherwise the bus fold over nsrc stereo sources,
// with the per-job results too when out != nullptr.  Frames [frame0, frame0 + frames_run) of the run.
hipError_t launch_lf_inv(int R, int C, const c32 *A, int nframes, int frame0, int frames_run, int njobs, int64_t base, int64_t hop, int64_t mpad,
                         int nsrc, float *bus0, float *bus1, int64_t tl, const double *vol, float *out, int64_t out_stride,
                         int64_t out_len, const c32 *tab, bool wide, hipStream_t s)
{
    if (frames_run <= 0) return hipSuccess;
    if (nframes > 65535 || njobs > 65535) return hipErrorInvalidValue;
    return lf_dispatch(R, C, [&](auto r, auto c) {
        return lf_inv_t<int(r), int(c)>(A, nframes, frame0, frames_run, njobs, base, hop, mpad, nsrc, bus0, bus1, tl, vol, out, out_stride, out_len,
                                        tab, wide, s);
    });
}

__global__ void k_add_prev(float *__restrict__ out, int64_t out_stride, const float *__restrict__ prev, int64_t prev_stride,
                           int64_t prev_len)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= prev_len) return;
    const int job = blockIdx.y;
    out[(int64_t)job * out_stride + i] += prev[(int64_t)job * prev_stride + i];
}

hipError_t launch_add_prev(float *out, int64_t out_stride, const float *prev, int64_t prev_stride, int64_t prev_len, int njobs,
                           hipStream_t s)
{
    if (prev_len <= 0 || njobs <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_add_prev, dim3((unsigned)((prev_len + 255) / 256), njobs), dim3(256), 0, s, out, out_stride, prev,
                       prev_stride, prev_len);
    return hipGetLastError();
}

}  // namespace wp
