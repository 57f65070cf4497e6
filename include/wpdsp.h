/*
 * wpdsp.h — C ABI of libwpdsp.so: the MI355X (gfx950) implementation of
 * wavepot/dsp's convolution-reverb and mix-buffers hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or framework
 * types.  The Node N-API addon (dsp_amd/napi/wpdsp_napi.cc), the JavaScript
 * shims (dsp_amd/js/) and the Python binding (dsp_amd/capi.py) are thin callers
 * of exactly these entry points.  Each declaration names the reference code it
 * stands in for (paths relative to the reference checkout).
 *
 * Conventions
 *   - Functions returning int: 0 on success, negative WP_ERR_* on failure;
 *     wp_last_error() gives the message for the calling thread.
 *   - "_dev" entry points take device pointers and a hipStream_t (as void*;
 *     NULL = the default stream) and are asynchronous on that stream.  The others
 *     take host pointers, copy in and out, and return when the result is in
 *     host memory.
 *   - Handles belong to the HIP device that was current when they were created.
 *     One handle must not be used from two threads at once; different handles
 *     may be.  A wp_ir / wp_conv / wp_reverb handle owns one workspace: its calls
 *     may move from stream to stream (a call on a new stream waits, on the
 *     device, for the handle's previous call), but are never concurrent.
 *   - Without a usable GPU every compute entry point fails with
 *     WP_ERR_NO_DEVICE.  There is no CPU fallback.
 */
#ifndef WPDSP_H
#define WPDSP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WP_OK 0
#define WP_ERR_INVALID (-1)     /* bad argument */
#define WP_ERR_RANGE (-2)       /* input longer than the FFT size: RangeError in the reference (lib/convolve.js:590) */
#define WP_ERR_FFT_SIZE (-3)    /* "FFT size must be a power of two and bigger than 1" (lib/convolve.js:8-9) */
#define WP_ERR_NO_DEVICE (-4)   /* no HIP device / runtime failure at init */
#define WP_ERR_HIP (-5)         /* a HIP call failed; see wp_last_error() */
#define WP_ERR_UNSUPPORTED (-6) /* valid in the reference, not implemented here; message says what */

typedef struct wp_ir wp_ir;         /* device-resident partitioned spectra of an impulse response */
typedef struct wp_conv wp_conv;     /* a batch of channel-jobs convolving with one wp_ir */
typedef struct wp_reverb wp_reverb; /* wp_conv + carried tail per job (block reverb) */
typedef struct wp_buf wp_buf;       /* device-resident sample buffer: channels x frames of f32 */
typedef struct wp_comm wp_comm;     /* one rank of a multi-GPU render: an RCCL communicator */

/* Number of visible HIP devices, or WP_ERR_NO_DEVICE.  Idempotent, thread-safe.
 * Also looks up, for the HIP device current at the call, the address of the
 * library's static workspace (see wp_convolve_dev_noalloc). */
int wp_init(void);
/* Message for the last failure on the calling thread ("" if none). */
const char *wp_last_error(void);

/* lib/convolve.js:521-530 nextPowerOfTwo (int32 bit smear; 0 -> 1). */
int32_t wp_next_pot(int32_t n);
/* lib/convolve.js:554-555, 565-566: FFT size the reference would use for block
 * length L and kernel length M; the interleaved spectrum it caches has 2x this
 * many floats. */
int32_t wp_fft_length(int64_t L, int64_t M);

/* ---- impulse response --------------------------------------------------- */
/* lib/convolve.js:553-562 fftProcessKernel(length, kernel), for nch kernels of
 * M samples each (host memory): builds the partition spectra on the device.
 * L is the block length the caller convolves with (c.buffer[0].length in
 * lib/impulse-convolve.js:12, impulse-convolve-stereo.js:14-15); it fixes the
 * result length L+M and the partition size.  NULL on failure.  Handles made from
 * identical samples and lengths share one copy of the spectra in device memory
 * (the kernel object travels between worker threads through the cache,
 * lib/impulse-convolve.js:9-13; each thread still gets a handle of its own). */
wp_ir *wp_ir_create(const float *const *ir, int nch, int64_t M, int64_t L);
/* Same with the IR already on the device: channel c starts at ir_dev + c*ch_stride. */
wp_ir *wp_ir_create_dev(const float *ir_dev, int nch, int64_t ch_stride, int64_t M, int64_t L, void *stream);
void wp_ir_destroy(wp_ir *ir);
/* which: 0 M, 1 L, 2 channels, 3 partition size B, 4 partitions P, 5 result length L+M,
 * 6 reference FFT length, 7 device bytes of spectra, 8 handles sharing them,
 * 9 frame length of the long-frame form's first run of frames (0: results of L+M samples use the
 * partitioned form), 10 runs of equal frames, 11 frame samples per job over all runs. */
int64_t wp_ir_info(const wp_ir *ir, int which);
/* Host only (no GPU needed, like the integer helpers): the cover by runs of equal frames that an
 * IR handle of nch channels would get for results of L+M samples (one go of lib/convolve.js:564-586
 * with input.length = L), under the current wp_tune settings.  Returns the number of runs (0: the
 * partitioned form) and writes, per run, frame length in samples, frames per job, samples a frame
 * advances by, first output sample into runs[4*i .. 4*i+3] (room for 3 runs). */
int wp_lf_cover(int64_t L, int64_t M, int nch, int64_t *runs /* 12 */);
/* Host only: of the `frames` frames (256 rows of `columns` = 2048 or 4096 points) that a batched one-pass call
 * (wp_conv_run_mix_dev: every go() result of lib/convolve.js:564-586 written and src/mix-buffers.js:49-52
 * folded in the same pass) runs of a plan's first run of frames, how many take bus-fold workgroups over tiles
 * of 32 columns instead of 16 on a device of `compute_units` CUs, under the current WP_TUNE_LF_WIDE setting.
 * Results do not depend on it. */
int wp_lf_wide_frames(int columns, int frames, int compute_units);

/* ---- convolution: lib/convolve.js:564-586 fftConvolution(...) -> go(input) - */
/* One go(input): ch picks the IR channel; in_len <= reference FFT length, else
 * WP_ERR_RANGE; out receives exactly L+M floats.  An input longer than L whose
 * convolution runs past the FFT length comes back wrapped, as the reference's
 * circular convolution returns it (the batched and block entry points below
 * answer WP_ERR_UNSUPPORTED for such inputs). */
int wp_convolve(wp_ir *ir, int ch, const float *in, int64_t in_len, float *out);
/* The same with input and result in device memory (in_len floats in, L+M out). */
int wp_convolve_dev(wp_ir *ir, int ch, const float *in_dev, int64_t in_len, float *out_dev, void *stream);

/* lib/convolve.js:541-551 directConvolution(input, kernel): out receives n+m-1
 * floats, each the reference's own chain (per output sample, input index
 * ascending: binary64 multiply, binary64 add, binary32 store), so the result is
 * bit-identical to the reference's.  Host pointers. */
int wp_direct_convolution(const float *input, int64_t n, const float *kernel, int64_t m, float *out);
/* The kernel object in the reference's own layout (lib/convolve.js:553-562,588-595):
 * the interleaved complex spectrum of the kernel zero-padded to N = wp_fft_length(L, M)
 * points, 2N floats.  Computed with binary64 butterflies and rounded once, so it
 * equals the reference's array to the reference's own rounding noise (the
 * reference rounds to binary32 after every radix-4 pass).  Host pointers. */
int wp_kernel_spectrum(const float *kernel, int64_t M, int64_t L, float *spectrum_out /* 2N */);
/* The way back: the first M samples of the inverse transform's real part.  Lets
 * fftConvolution accept a spectrum that the reference's fftProcessKernel produced
 * (lib/convolve.js:575-579 uses any Float32Array(2N) as the kernel). */
int wp_kernel_from_spectrum(const float *spectrum /* 2N */, int64_t N, int64_t M, float *kernel_out /* M */);

/* No counterpart in the reference.  wp_convolve_dev for callers that cannot
 * allocate device memory when they call (code under HIP stream capture, tight
 * serving loops, harnesses that forbid hipMalloc): the block spectra live in a
 * static device array of WP_NOALLOC_WORKSPACE_BYTES that the library's code object
 * carries (one copy per GPU, held until the process ends), and the call makes no
 * hipMalloc/hipFree or other allocating call.  It always runs the partitioned form
 * (same kernels and geometry as a wp_convolve_dev plan in that form, so the result is
 * bit-identical to such a plan's); for inputs many times longer than the impulse
 * response wp_convolve_dev itself may choose the long-frame form, whose results differ
 * from these in the last bits (both within 1e-5 of the reference's).  Limits: the result
 * length L+M must satisfy ceil((L+M)/B)*B <= WP_NOALLOC_WORKSPACE_BYTES/16 complex
 * points (B = wp_ir_info(ir, 3); with B = 4096 that is L+M <= 524288 samples, 10.9 s
 * at 48 kHz), else WP_ERR_RANGE; inputs that wrap in the reference's FFT are
 * WP_ERR_UNSUPPORTED (use wp_convolve_dev).  Call wp_init() (or this function once)
 * on the device before capturing.  All calls on one GPU share the workspace: calls
 * that overlap in time on one GPU (two streams, two threads) race; calls on one
 * stream are ordered and safe. */
#define WP_NOALLOC_WORKSPACE_BYTES (8 * 1024 * 1024)
int wp_convolve_dev_noalloc(wp_ir *ir, int ch, const float *in_dev, int64_t in_len, float *out_dev, void *stream);

/* Batched: njobs independent go() calls sharing one launch sequence.  job_ch[j]
 * (host array) is the IR channel job j uses. */
wp_conv *wp_conv_create(wp_ir *ir, int njobs, const int32_t *job_ch);
void wp_conv_destroy(wp_conv *c);
/* Job j reads in_dev + j*in_stride (in_len samples) and writes L+M samples at
 * out_dev + j*out_stride.  If prev_dev is non-NULL, prev_dev[j*prev_stride + i]
 * is added to output sample i for i < prev_len (lib/impulse-reverb.js:14-16). */
int wp_conv_run_dev(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, float *out_dev,
                    int64_t out_stride, const float *prev_dev, int64_t prev_stride, int64_t prev_len,
                    void *stream);
/* Fused reverb + mix for a batch of stereo tracks (jobs 2s, 2s+1 = left, right of
 * source s).  Produces what go() per channel (lib/convolve.js:570-583) followed by
 *   mixBuffers(tgt, [[y_0,y_1],1,vol[0],0], [[y_2,y_3],1,vol[1],0], ...)
 * (src/mix-buffers.js:49-52) produces in tgt_dev[0..1] (tl <= L+M frames, updated in
 * place), bit-identically, but the per-track results are folded into the bus
 * inside the inverse-transform kernel and never written to memory.  vol is a
 * host array of njobs/2 gains. */
int wp_conv_mix_run_dev(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, float *const *tgt_dev,
                        int64_t tl, const double *vol, void *stream);
/* The same in one pass WITH the per-track results: out_dev receives every job's
 * L+M samples exactly as wp_conv_run_dev writes them, and the bus is folded from
 * those samples inside the same kernel (the results are not read back).  Equals
 * wp_conv_run_dev followed by wp_mix_dev, bit for bit, on both outputs. */
int wp_conv_run_mix_dev(wp_conv *c, const float *in_dev, int64_t in_stride, int64_t in_len, float *out_dev,
                        int64_t out_stride, float *const *tgt_dev, int64_t tl, const double *vol, void *stream);
/* Device bytes of workspace a wp_conv holds. */
int64_t wp_conv_workspace_bytes(const wp_conv *c);
/* Measurement hook (no reference counterpart): when enabled, the wp_conv_*run*_dev
 * calls bracket each of their three passes (forward transforms, spectral product,
 * inverse transforms) with HIP events on the caller's stream.  wp_conv_profile_read
 * waits for the last run and returns its three durations in milliseconds.
 * wp_conv_profile_sum waits for every profiled run since the previous call of it,
 * returns the three durations added up over those runs and how many they were, and
 * starts a new sum: the runs can be queued back to back without a wait between them
 * (the events rotate through 32 sets; a set that comes round again is added up first). */
int wp_conv_profile(wp_conv *c, int enable);
int wp_conv_profile_read(wp_conv *c, float *ms3);
int wp_conv_profile_sum(wp_conv *c, double *ms3_sum, int *runs);
/* Geometry of a plan: which = 0 jobs, 1 output blocks T, 2 partition size B,
 * 3 partitions P (padded), 4 blocks per MAC time chunk, 5 whether the small-batch
 * (gather) form of the spectral multiply-accumulate is in use, 6 whether the plan
 * runs the long-frame form (then, for its first run of equal frames: 7 frame length in
 * samples, 8 frames per job, 9 samples a frame advances by; 10 number of runs, 11 frame
 * samples per job over all runs; 16+4i.. 19+4i for run i: frame length, frames, advance,
 * first output sample). */
int64_t wp_conv_info(const wp_conv *c, int which);

/* ---- block reverb: lib/impulse-reverb.js:10-21, impulse-reverb-stereo.js:11-30 */
/* Index bookkeeping of one reverb block, in samples relative to the start of
 * the (L+M)-sample result buffer `curr`:
 *   ret  = curr.subarray(offset, offset+len)      (impulse-reverb.js:20)
 *   prev = curr.subarray(-tail), tail = (L+M-offset)-len   (:17-18), with
 *          TypedArray.subarray's clamping (tail == 0 keeps everything, tail < 0
 *          drops |tail| samples from the front). */
typedef struct wp_reverb_view {
    int64_t curr_len; /* L+M */
    int64_t ret_off, ret_len;
    int64_t prev_off, prev_len;
} wp_reverb_view;
/* Pure integer helper (no GPU): the view for block length len. */
int wp_reverb_view_for(int64_t L, int64_t M, int64_t offset, int64_t len, wp_reverb_view *v);

/* njobs channel-jobs (mono reverb: 1, stereo: 2 with job_ch = {0,1}; a batch of
 * tracks: 2*tracks), each with its own carried tail. */
wp_reverb *wp_reverb_create(wp_ir *ir, int njobs, const int32_t *job_ch, int64_t offset);
void wp_reverb_destroy(wp_reverb *r);
/* One block for every job, host buffers: in[j] has in_len samples, out[j]
 * receives the whole curr buffer (L+M floats, tail already added); *v tells
 * which parts the reference returns and carries. */
int wp_reverb_block(wp_reverb *r, const float *const *in, int64_t in_len, float *const *out, wp_reverb_view *v);
/* Same on the device: job j reads in_dev + j*in_stride; *curr_dev is set to the
 * library-owned result (job j at *curr_dev + j*(*curr_stride)), valid until the
 * next-but-one block. */
int wp_reverb_block_dev(wp_reverb *r, const float *in_dev, int64_t in_stride, int64_t in_len, float **curr_dev,
                        int64_t *curr_stride, wp_reverb_view *v, void *stream);
/* Same, but curr goes into the caller's buffer (job j at curr_dev + j*curr_stride,
 * L+M floats each) -- the counterpart of go() returning a fresh array per block
 * (lib/impulse-reverb.js:12).  The carried tail is then a view INTO that buffer
 * (prev = curr.subarray(-tail), :18): the caller keeps it alive and unmodified
 * until the next block has run. */
int wp_reverb_block_to(wp_reverb *r, const float *in_dev, int64_t in_stride, int64_t in_len, float *curr_dev,
                       int64_t curr_stride, wp_reverb_view *v, void *stream);
/* Carried tail in/out (the value lib/impulse-reverb.js:6,19 keeps under
 * 'prev:'+id+url+n): prev[j] has prev_len samples. */
int wp_reverb_set_tail(wp_reverb *r, const float *const *prev, int64_t prev_len);
/* Tail taken from device memory without a copy (job j at prev_dev + j*prev_stride);
 * the caller keeps it alive until the next block has run. */
int wp_reverb_set_tail_dev(wp_reverb *r, const float *prev_dev, int64_t prev_stride, int64_t prev_len);
int64_t wp_reverb_tail_len(const wp_reverb *r);
int wp_reverb_get_tail(wp_reverb *r, float *const *prev_out);

/* ---- mix: src/mix-buffers.js:1-73 ---------------------------------------- */
/* One decoded source (src/mix-buffers.js:5-14).  target[c][x] +=
 * f(src[(x+o) % sl % rl]) * vol, evaluated per source in order with binary64
 * multiply and add and a binary32 store, exactly as the reference does. */
typedef struct wp_src {
    const float *ch[2];    /* channel pointers; ch[1] ignored when nch == 1 */
    int32_t nch;           /* 1 or 2; anything else is skipped like the reference does */
    int32_t flags;         /* reserved, 0 */
    int64_t rl;            /* real length: source[0].length (:11,:13) */
    int64_t sl;            /* (rl*length)|0 for the tuple form (:7), rl for the plain form */
    double vol;            /* source[2] ?? 1 (:8) */
    const double *vol_vec; /* if non-NULL, gain per target frame: _vol(x) (:21), tl entries */
    int64_t o;             /* source[3] ?? 0 (:9) */
} wp_src;
/* src/mix-buffers.js:7: sl = (rl*length)|0 with JavaScript ToInt32 semantics. */
int32_t wp_mix_sl(int64_t rl, double length);
/* Host buffers: tgt[0..tch) of tl floats are updated in place.  Arrays that share
 * memory (a source that is the target itself, a view into it or one that reaches beyond
 * it) share it on the device too, and such a source is read sample by sample as the loop
 * updates the target, like the reference's (src/mix-buffers.js:4, 20-67: mixBuffers(t, a, t)
 * leaves 2(t+a)). */
int wp_mix(float *const *tgt, int tch, int64_t tl, const wp_src *srcs, int64_t nsrc);
/* Device buffers: every pointer inside tgt_dev/srcs is a device pointer; the
 * srcs array itself is in host memory unless srcs_on_device != 0.  The target is
 * updated in place.  A source that overlaps it in memory is folded frame by frame in
 * the reference's order by one lane (exact for any overlap, about a microsecond per
 * frame); the sources before and after it take the tiled kernel.  A source table
 * that lives on the device is taken to be free of overlaps. */
int wp_mix_dev(float *const *tgt_dev, int tch, int64_t tl, const wp_src *srcs, int64_t nsrc, int srcs_on_device,
               void *stream);

/* ---- multi-GPU: the master bus of a track-sharded render ---------------------- */
/* Tracks are independent buffers in the reference (each c.src / c.render renders
 * into its own, src/context.js:145-162) and the only coupling is their sum into
 * the master buffer (src/mix-buffers.js:1-73, src/mix.js:15-27).  With one process
 * per GPU, rank r reverbs and mixes its share of the tracks into a rank-local
 * partial bus; ONE reduce combines the partial buses.  That reduce is RCCL over
 * xGMI; RCCL is opened (dlopen) at the first wp_comm_* call, from the directory of
 * the HIP runtime the process uses.  The cross-rank sum adds the partial buses in
 * RCCL's order, so the N-rank bus matches the sequential mix to rounding (<= 1e-5
 * of full scale for buses within +-1), not bit for bit; one rank returns its bus
 * unchanged. */
#define WP_COMM_ID_BYTES 128
/* Rank 0 makes the id (ncclGetUniqueId) and hands its bytes to the other ranks by
 * whatever channel the host has (a file, a socket, a worker message). */
int wp_comm_unique_id(unsigned char *id_out /* WP_COMM_ID_BYTES */);
/* Every rank, with its GPU current (hipSetDevice): joins the communicator; returns
 * when all nranks ranks have.  NULL on failure. */
wp_comm *wp_comm_create(const unsigned char *id, int nranks, int rank);
void wp_comm_destroy(wp_comm *c);
/* which: 0 ranks, 1 this rank, 2 device. */
int64_t wp_comm_info(const wp_comm *c, int which);
/* Path of the RCCL library in use ("" before the first wp_comm_* call). */
const char *wp_comm_library(void);
/* Not collective: WP_OK when this process has a GPU and RCCL can be opened, so that a
 * host can agree on readiness among its ranks BEFORE any of them enters
 * wp_comm_create (which waits for all of them); an error code otherwise.  The
 * reference has no counterpart (its workers share memory, src/mix-worker-thread.js:14-16). */
int wp_comm_available(void);
/* recv[i] = sum over ranks of send[i], i < count, on rank `root` (recv_dev may be
 * NULL elsewhere and may equal send_dev); asynchronous on `stream`.  A stereo bus
 * of F frames stored as [2][F] is one call with count = 2*F. */
int wp_bus_reduce(wp_comm *c, const float *send_dev, float *recv_dev, int64_t count, int root, void *stream);
/* The same with the sum delivered to every rank. */
int wp_bus_allreduce(wp_comm *c, const float *send_dev, float *recv_dev, int64_t count, void *stream);

/* ---- page-locked host memory -------------------------------------------------- */
/* The reference's sample buffers are long-lived shared arrays handed round bar after bar
 * (lib/shared-array-buffer.js:1-8, src/buffer-service.js:26-43, c.buffer of
 * src/context.js).  When every host array of a call to the host-pointer entry points
 * (wp_convolve, wp_reverb_block, wp_mix) is page-locked, the call's transfers are DMA
 * straight from and to the arrays, queued with its kernels on a stream of the calling
 * thread and awaited once; with a pageable array among them the call uses blocking
 * copies through the runtime's bounce buffers.  wp_host_alloc returns page-locked
 * memory (NULL on failure); wp_host_pin locks a range the caller owns (unlock it before
 * freeing; a platform may refuse, e.g. inside some containers: WP_ERR_HIP, and the range
 * simply stays pageable).  Results are the same either way. */
void *wp_host_alloc(size_t bytes);
void wp_host_free(void *p);
int wp_host_pin(void *p, size_t bytes);
int wp_host_unpin(void *p);

/* ---- device-resident sample buffers ---------------------------------------- */
/* The reference keeps samples in arrays of Float32Array channels over
 * SharedArrayBuffers that every worker sees (src/buffer-service.js:8-27,
 * src/mix-worker-thread.js:14-16).  A wp_buf is that array in HBM: nch channels of
 * `frames` floats, zero-filled on creation like a fresh typed array, shared
 * between the threads of the process by id, reference-counted.  Channel c starts
 * at wp_buf_channel(b, c); the "_dev" entry points above take such pointers, so
 * a bar can go render -> reverb -> mix without leaving the device. */
wp_buf *wp_buf_create(int nch, int64_t frames);
/* Process-wide id (> 0) another thread can open; 0 for NULL. */
uint64_t wp_buf_id(const wp_buf *b);
/* One more reference to a live buffer, or NULL if the id names none. */
wp_buf *wp_buf_open(uint64_t id);
/* Drop one reference; the memory is recycled when the last one goes: the next
 * wp_buf_create of that size may hand it out again, zero-filled on the default
 * stream.  Work queued on a NON-BLOCKING stream that still uses the buffer is not
 * waited for: release a buffer only after such work has completed. */
void wp_buf_release(wp_buf *b);
/* which: 0 channels, 1 frames, 2 floats between channels, 3 device, 4 bytes. */
int64_t wp_buf_info(const wp_buf *b, int which);
float *wp_buf_channel(const wp_buf *b, int ch);
/* channel.set(hostArray, dst_off): the worker's copy-back (src/mix-worker-thread.js:23-24)
 * as an upload.  A span past the end is WP_ERR_RANGE, as TypedArray.set throws. */
int wp_buf_write(wp_buf *b, int ch, int64_t dst_off, const float *src_host, int64_t n);
/* Download n frames of one channel. */
int wp_buf_read(const wp_buf *b, int ch, int64_t src_off, float *dst_host, int64_t n);
/* dst.set(src) between device channels (src/render.js:45 buffer[1].set(buffer[0]);
 * overlapping ranges copy as if through a temporary). */
int wp_buf_copy(wp_buf *dst, int dch, int64_t dst_off, const wp_buf *src, int sch, int64_t src_off, int64_t n,
                void *stream);
/* channel.fill(value, off, off+n) (src/context.js:140-143 zero()). */
int wp_buf_fill(wp_buf *b, int ch, int64_t off, int64_t n, float value, void *stream);

/* ---- integer helpers ------------------------------------------------------ */
/* lib/ring-buffer.js:10-22,33-50: element a proxy access at `key` touches. */
int64_t wp_ring_pos(int64_t needle, int64_t key, int64_t length);
/* lib/ring-buffer.js:35-39: needle after a write at key 0. */
int64_t wp_ring_advance(int64_t needle, int64_t length);
/* -1 for an empty array (length <= 0), where the reference's wrap has no element to name. */
/* src/loop-node.js:145-147 beatRate = Math.round(sampleRate*(60/bpm)); :48-50
 * bufferSize = 4*beatRate.  -1 when the rate is not finite (bpm 0). */
int64_t wp_beat_rate(double sample_rate, double bpm);
/* src/loop-node.js:9-11 the bpm getter: parseFloat((60*(sampleRate/beatRate)).toFixed(6)). */
double wp_bpm_from_beat_rate(double sample_rate, int64_t beat_rate);

/* ---- tuning knobs for tests and measurements (no reference counterpart) ----- */
/* Process-wide overrides of choices the library otherwise makes itself; value 0
 * restores the library's choice.  They affect handles created afterwards. */
#define WP_TUNE_PARTITION 0 /* partition size B: 1024, 2048, 4096 or 8192 */
#define WP_TUNE_MAC_GROUP 1 /* partitions per multiply-accumulate pass: 4, 8, 12, 16, 24 */
#define WP_TUNE_MAC_FORM 2  /* 1 streaming delay-line kernel, 2 small-batch gather kernel */
#define WP_TUNE_CONV_FORM 5 /* convolution form of IR handles created afterwards: 1 partitioned, 2 long-frame where the sizes allow */
#define WP_TUNE_LF_RUNS 6   /* long-frame form: at most this many runs of equal frames per result (1..3; 1 = one frame size) */
#define WP_TUNE_MID_FORM 7  /* long-frame row pass: 1 one job per workgroup, 2 eight (the default for batches; takes effect at once) */
#define WP_TUNE_LF_SPECTRA 8 /* long-frame kernel spectra of IR handles created afterwards: 1 plain spectra, 2 product coefficients when the jobs share few kernels (default) */
#define WP_TUNE_LF_MAX_FRAME 9  /* long-frame form: frames of at most 2^n samples (17..21; 0 = no limit) for IR handles created afterwards */
#define WP_TUNE_LF_MIN_FRAME 10 /* long-frame form: frames of at least 2^n samples (17..21; 0 = no limit) */
#define WP_TUNE_LF_COST 11 /* long-frame form: 1 = choose the cover by its bytes alone (0: measured per-shape kernel rates weigh in) */
#define WP_TUNE_LF_SPLIT 12 /* long-frame form, one-pass results + bus: 1 = every run of frames folds the bus inside its inverse pass, 2 = every run after the first takes the plain pass + the mix kernel over its range, 0 = the later runs fold in their pass while the chip has room for their workgroups beside the first run's (takes effect at once) */
#define WP_TUNE_LF_WIDE 13 /* long-frame form, results + bus in one pass, frames of 256 rows: 1 = bus-fold chains over tiles of 16 columns only, 2 = of 32 columns only, 0 = the split with the shortest schedule (takes effect at once) */
#define WP_TUNE_FWD_FORM 4  /* forward transform: 1 one segment per workgroup, 2 two (the default; takes effect at once) */
#define WP_TUNE_POOL_MB 3   /* cap of the wp_buf recycling pool in MiB (default 2048; 0 disables it and the 64 MiB slabs that buffers of up to 8 MiB are carved from when the pool has none of their size; carved memory stays with the process) */
int wp_tune(int what, int64_t value);

/* ---- synthetic inputs for tests and benchmarks (no reference counterpart) --- */
/* out[i] = ((fmix32((start+i) ^ seed*0x9E3779B9) >> 8) - 2^23) / 2^23, on the device. */
int wp_synth_noise_dev(float *out_dev, int64_t n, uint32_t seed, int64_t start, void *stream);
/* Decaying-noise IR, L1-normalised, computed on the host with IEEE basic
 * operations only (bit-reproducible everywhere). */
int wp_synth_ir(float *out_host, int64_t M, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* WPDSP_H */
