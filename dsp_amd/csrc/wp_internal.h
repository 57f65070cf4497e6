// wp_internal.h — declarations shared by the kernel translation units and the
// C-ABI layer (not part of the public interface; see include/wpdsp.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "wp_fft.h"
#include "wp_lf_layout.h"  // LF_C, LF_C_BIG and the layout of a frame in the long-frame workspace
#include "../../include/wpdsp.h"

namespace wp {

// dynamic LDS of an FFT kernel: the exchange buffer, plus the twiddle tables for the
// kernels that stage them (TABLES)
template <int B, bool TABLES = true>
constexpr size_t fft_lds_bytes()
{
    return sizeof(c32) * (size_t)(lds_points<B>() + (TABLES ? tw_table_points<B>() : 0));
}

// device table block for one partition size B:
//   [0,256) t256 | [256,512) tlo | [512, 512+B/256) thi
//   | thalf[1024] = exp(-i*pi*j/B) | wbin[B] = exp(-i*pi*(k+1/2)/B)
// (thalf is indexed by j < max(B/PT, B/last radix): at most 1024 for the plans in wp_fft.h)
__host__ __device__ inline int table_thalf_offset(int B) { return 512 + B / 256; }
__host__ __device__ inline int table_wsplit_offset(int B) { return 512 + B / 256 + 1024; }
inline int table_total_points(int B) { return 512 + B / 256 + 1024 + B; }

hipError_t launch_fft_fwd(int B, bool irmode, const float *in, int64_t in_stride, int64_t in_len, c32 *Z, int T,
                          int nblocks, int njobs, const c32 *tables, hipStream_t s);
void set_fwd_pair(bool on);  // forward transform form: two segments per workgroup (default) or one
hipError_t launch_fft_inv(int B, const c32 *W, int T, int njobs, float *out, int64_t out_stride, int64_t out_len,
                          const float *prev, int64_t prev_stride, int64_t prev_len, const c32 *tables, hipStream_t s);
hipError_t launch_hpack(const c32 *Zir, c32 *Hc, const c32 *wbin, int B, int P, int Pp, int nch, hipStream_t s);
// out[i] = lin[i] + lin[i+N] (second term while i+N < lin_len), i < out_len
hipError_t launch_wrap_add(const float *lin, int64_t lin_len, int64_t N, float *out, int64_t out_len, hipStream_t s);
// job_irch: device table of the IR channel each job uses, or nullptr when every job uses channel irch_all
hipError_t launch_mac_gather(const c32 *Z, c32 *W, const c32 *Hc, const c32 *wbin, const int32_t *job_irch,
                             int irch_all, int B, int T, int Tout, int Tin, int Pp, int njobs, hipStream_t s);
hipError_t launch_mac(int PG, const c32 *Z, c32 *W, const c32 *Hc, const c32 *wbin, const int32_t *job_irch,
                      int irch_all, int B, int T, int Tout, int Tin, int Pp, int njobs, int tchunk, hipStream_t s);
hipError_t launch_fft_inv_mix(int B, const c32 *W, int T, int nblocks, int nsrc, float *bus0, float *bus1, int64_t tl,
                              const double *vol, float *out, int64_t out_stride, int64_t out_len, const c32 *tables,
                              hipStream_t s);

hipError_t launch_direct_conv(const float *x, int64_t n, const float *h, int64_t m, float *out, hipStream_t s);
// N-point complex transform in binary64 over global memory (work0/work1: 16*N bytes each)
hipError_t launch_big_fft(const float *src, int64_t n_real, int64_t N, int inverse, void *work0, void *work1, float *dst,
                          int64_t n_out, hipStream_t s);

// ---- long-frame convolution (wp_longframe_kernels.hip) ----
// A frame of 2*Nc real samples = Nc = R*C complex points: R = 32, 64, 128 or 256 rows of C = LF_C points, or 256 rows of
// LF_C_BIG points (2^17 ... 2^21 samples).  A job's result is covered by up to LF_MAX_SEG runs of equal frames (largest
// first); `base` is the first output sample of a run.
constexpr int LF_MAX_SEG = 3;
// device table block of one (R, C) (c32 entries):
//   [LF_T256, +256)  exp(-2*pi*i*a/256)            [LF_ROW, +256)   exp(-i*pi*r/R), r < R
//   [LF_COL, +4096)  exp(-i*pi*c/Nc), c < C        [LF_FLO, +512)   exp(-2*pi*i*b/Nc), b < 512
//   [LF_FHI, +2048)  exp(-2*pi*i*a*512/Nc)         [LF_WA, +256)    exp(-i*pi*(k1+1/2)/Nc), k1 < R
//   [LF_WB, +4096)   exp(-i*pi*k2/C), k2 < C
constexpr int LF_T256 = 0, LF_ROW = 256, LF_COL = 512, LF_FLO = 4608, LF_FHI = 5120, LF_WA = 7168, LF_WB = 7424,
              LF_TABLE_POINTS = 11520;
bool lf_shape_ok(int R, int C);  // a frame shape the kernels are built for
// nframes: frames per job in A (the stride); frames_run <= nframes: how many of them this launch handles
hipError_t launch_lf_fwd(int R, int C, const float *in, int64_t in_stride, int64_t in_len, c32 *A, int nframes, int frames_run,
                         int njobs, int64_t base, int64_t hop, int64_t mpad, const c32 *tab, hipStream_t s);
void set_mid_jobs(int jobs);  // WP_TUNE_MID_FORM
// tab_row: the table block of the C-point plan (row transforms)
// ab: Hs holds (alpha, beta) coefficient pairs, 16 bytes per bin (kernel spectra shared by the jobs), instead of spectra
hipError_t launch_lf_mid(bool hprep, bool ab, int R, int C, c32 *A, c32 *Hs, const int32_t *job_irch, int irch_all, int nframes,
                         int frames_run, int njobs, const c32 *tab, const c32 *tab_row, hipStream_t s);
// bus0 == nullptr: plain results for every job; otherwise the bus fold over nsrc stereo sources (jobs 2s, 2s+1),
// with the per-job results too when out != nullptr
hipError_t launch_lf_inv(int R, int C, const c32 *A, int nframes, int frame0, int frames_run, int njobs, int64_t base, int64_t hop, int64_t mpad,
                         int nsrc, float *bus0, float *bus1, int64_t tl, const double *vol, float *out, int64_t out_stride,
                         int64_t out_len, const c32 *tab, bool wide, hipStream_t s);
// out[job][i] += prev[job][i], i < prev_len (the carried reverb tail, lib/impulse-reverb.js:14-16)
hipError_t launch_add_prev(float *out, int64_t out_stride, const float *prev, int64_t prev_stride, int64_t prev_len, int njobs,
                           hipStream_t s);

size_t mix_scratch_bytes(int64_t nsrc);  // bytes of device scratch launch_mix needs
hipError_t launch_mix(float *t0, float *t1, int tch, int64_t tl, const wp_src *srcs_dev, void *scratch,
                      int64_t nsrc, hipStream_t s);
// one source that overlaps the target in memory, frame by frame in the reference's order (device pointers in `src`)
hipError_t launch_mix_serial(float *t0, float *t1, int tch, int64_t tl, const wp_src &src, hipStream_t s);
hipError_t launch_synth_noise(float *out, int64_t n, uint32_t seed, int64_t start, hipStream_t s);

}  // namespace wp
