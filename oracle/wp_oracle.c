/*
 * wp_oracle.c — CPU restatement of wavepot/dsp's convolution + mix hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or
 * calls this file; it exists so that tests/, the smoke check and the bench's
 * cpu_baseline leg have a CPU answer to compare the HIP path against.
 *
 * Parity status: PINNED.  Every routine here is checked bit-for-bit against
 * vectors produced by running the reference's own JavaScript under Node
 * (oracle/gen_golden.mjs -> tests/golden/, tests/test_oracle_golden.py).
 *
 * Arithmetic contract (what makes this bit-identical to V8 running the
 * reference): typed-array loads widen f32 -> f64, every JS arithmetic operator
 * is one IEEE-754 binary64 operation (build with -ffp-contract=off, no
 * fast-math), every typed-array store rounds f64 -> f32 (round-to-nearest-even).
 *
 * Each function cites the reference lines it follows (paths relative to the
 * reference checkout).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------ */
/* lib/convolve.js:521-530  nextPowerOfTwo (32-bit bit smear, 0 -> 1)        */
int32_t wpo_next_pot(int32_t n)
{
    if (n == 0) return 1;
    uint32_t v = (uint32_t)(n - 1);
    int32_t s = (int32_t)v;
    s |= s >> 1;
    s |= s >> 2;
    s |= s >> 4;
    s |= s >> 8;
    s |= s >> 16;
    return s + 1;
}

/* ------------------------------------------------------------------------ */
/* lib/convolve.js:4-45  FFT constructor state                              */
typedef struct {
    int32_t size;   /* N (complex points) */
    int32_t csize;  /* 2N floats */
    int32_t width;  /* :30 */
    float *table;   /* :14-19, 2N floats: cos, -sin */
    int32_t *bitrev;/* :33-40, only the half the transform ever reads */
    int32_t nbitrev;
} wpo_fft;

void wpo_fft_free(wpo_fft *f)
{
    if (!f) return;
    free(f->table);
    free(f->bitrev);
    free(f);
}

wpo_fft *wpo_fft_new(int32_t size)
{
    if (size <= 1 || (size & (size - 1)) != 0) return NULL; /* :8-9 throws */
    wpo_fft *f = (wpo_fft *)calloc(1, sizeof(*f));
    f->size = size;
    f->csize = size << 1;
    f->table = (float *)malloc(sizeof(float) * (size_t)f->csize);
    for (int32_t i = 0; i < f->csize; i += 2) {            /* :15-19 */
        const double angle = M_PI * (double)i / (double)size;
        f->table[i] = (float)cos(angle);
        f->table[i + 1] = (float)(-sin(angle));
    }
    int32_t power = 0;                                        /* :23-25 */
    for (int32_t t = 1; size > t; t <<= 1) power++;
    f->width = (power % 2 == 0) ? power - 1 : power;         /* :30 */
    /* :33-40.  The reference fills 1<<width entries, but the transform only
     * indexes the first 1<<(width-1) of them (csize/len leaves); the upper
     * half receives `<< -1` garbage that is never read.  For the lower half
     * the out-of-range shift contributes 0, so plain digit reversal it is. */
    f->nbitrev = 1 << (f->width - 1);
    f->bitrev = (int32_t *)malloc(sizeof(int32_t) * (size_t)f->nbitrev);
    for (int32_t j = 0; j < f->nbitrev; j++) {
        int32_t r = 0;
        for (int32_t shift = 0; shift < f->width; shift += 2) {
            int32_t rev = f->width - shift - 2;
            if (rev >= 0) r |= ((j >> shift) & 3) << rev;
        }
        f->bitrev[j] = r;
    }
    return f;
}

const float *wpo_fft_table(const wpo_fft *f) { return f->table; }
const int32_t *wpo_fft_bitrev(const wpo_fft *f) { return f->bitrev; }
int32_t wpo_fft_nbitrev(const wpo_fft *f) { return f->nbitrev; }
int32_t wpo_fft_width(const wpo_fft *f) { return f->width; }

/* lib/convolve.js:238-257 */
static void single_transform2(float *out, const float *data, int32_t outOff,
                              int32_t off, int32_t step)
{
    const double evenR = data[off];
    const double evenI = data[off + 1];
    const double oddR = data[off + step];
    const double oddI = data[off + step + 1];
    out[outOff] = (float)(evenR + oddR);
    out[outOff + 1] = (float)(evenI + oddI);
    out[outOff + 2] = (float)(evenR - oddR);
    out[outOff + 3] = (float)(evenI - oddI);
}

/* lib/convolve.js:262-311 */
static void single_transform4(float *out, const float *data, int32_t outOff,
                              int32_t off, int32_t step, double inv)
{
    const int32_t step2 = step * 2, step3 = step * 3;
    const double Ar = data[off], Ai = data[off + 1];
    const double Br = data[off + step], Bi = data[off + step + 1];
    const double Cr = data[off + step2], Ci = data[off + step2 + 1];
    const double Dr = data[off + step3], Di = data[off + step3 + 1];
    const double T0r = Ar + Cr, T0i = Ai + Ci;
    const double T1r = Ar - Cr, T1i = Ai - Ci;
    const double T2r = Br + Dr, T2i = Bi + Di;
    const double T3r = inv * (Br - Dr), T3i = inv * (Bi - Di);
    out[outOff] = (float)(T0r + T2r);
    out[outOff + 1] = (float)(T0i + T2i);
    out[outOff + 2] = (float)(T1r + T3i);
    out[outOff + 3] = (float)(T1i - T3r);
    out[outOff + 4] = (float)(T0r - T2r);
    out[outOff + 5] = (float)(T0i - T2i);
    out[outOff + 6] = (float)(T1r - T3i);
    out[outOff + 7] = (float)(T1i + T3r);
}

/* lib/convolve.js:128-233  _transform4 (out-of-place: out != data) */
static void transform4(const wpo_fft *f, float *out, const float *data, int inverse)
{
    const int32_t size = f->csize;
    const int32_t width = f->width;
    int32_t step = 1 << width;
    int32_t len = (size / step) << 1;
    const double inv = inverse ? -1.0 : 1.0;
    int32_t outOff, t;
    if (len == 4) {
        for (outOff = 0, t = 0; outOff < size; outOff += len, t++)
            single_transform2(out, data, outOff, f->bitrev[t], step);
    } else {
        for (outOff = 0, t = 0; outOff < size; outOff += len, t++)
            single_transform4(out, data, outOff, f->bitrev[t], step, inv);
    }
    const float *table = f->table;
    for (step >>= 2; step >= 2; step >>= 2) {
        len = (size / step) << 1;
        const int32_t quarterLen = len >> 2;
        for (outOff = 0; outOff < size; outOff += len) {
            const int32_t limit = outOff + quarterLen;
            for (int32_t i = outOff, k = 0; i < limit; i += 2, k += step) {
                const int32_t A = i, B = A + quarterLen, C = B + quarterLen,
                              D = C + quarterLen;
                const double Ar = out[A], Ai = out[A + 1];
                const double Br = out[B], Bi = out[B + 1];
                const double Cr = out[C], Ci = out[C + 1];
                const double Dr = out[D], Di = out[D + 1];
                const double MAr = Ar, MAi = Ai;
                const double tableBr = table[k];
                const double tableBi = inv * (double)table[k + 1];
                const double MBr = Br * tableBr - Bi * tableBi;
                const double MBi = Br * tableBi + Bi * tableBr;
                const double tableCr = table[2 * k];
                const double tableCi = inv * (double)table[2 * k + 1];
                const double MCr = Cr * tableCr - Ci * tableCi;
                const double MCi = Cr * tableCi + Ci * tableCr;
                const double tableDr = table[3 * k];
                const double tableDi = inv * (double)table[3 * k + 1];
                const double MDr = Dr * tableDr - Di * tableDi;
                const double MDi = Dr * tableDi + Di * tableDr;
                const double T0r = MAr + MCr, T0i = MAi + MCi;
                const double T1r = MAr - MCr, T1i = MAi - MCi;
                const double T2r = MBr + MDr, T2i = MBi + MDi;
                const double T3r = inv * (MBr - MDr), T3i = inv * (MBi - MDi);
                out[A] = (float)(T0r + T2r);
                out[A + 1] = (float)(T0i + T2i);
                out[B] = (float)(T1r + T3i);
                out[B + 1] = (float)(T1i - T3r);
                out[C] = (float)(T0r - T2r);
                out[C + 1] = (float)(T0i - T2i);
                out[D] = (float)(T1r - T3i);
                out[D + 1] = (float)(T1i + T3r);
            }
        }
    }
}

/* lib/convolve.js:86-96 */
void wpo_fft_transform(const wpo_fft *f, float *out, const float *data)
{
    transform4(f, out, data, 0);
}

/* lib/convolve.js:110-122 (the /size pass is a separate f32 sweep) */
void wpo_fft_inverse(const wpo_fft *f, float *out, const float *data)
{
    transform4(f, out, data, 1);
    const double size = (double)f->size;
    for (int32_t i = 0; i < f->csize; i++) out[i] = (float)((double)out[i] / size);
}

/* lib/convolve.js:588-595  createPaddedFFt: zero-pad to N, real -> interleaved
 * complex (imaginary parts stay 0), forward transform.  `out` holds 2N floats.
 * Returns -1 where the reference's input.set(data) would throw RangeError. */
static int padded_fft(const wpo_fft *f, const float *data, int64_t n, float *out)
{
    if (n > f->size) return -1;
    float *cplx = (float *)calloc((size_t)f->csize, sizeof(float));
    for (int64_t i = 0; i < n; i++) cplx[2 * i] = data[i];
    transform4(f, out, cplx, 0);
    free(cplx);
    return 0;
}

/* lib/convolve.js:553-562  fftProcessKernel(length, kernel).
 * out_spectrum must hold 2*nextPOT(length+M) floats. Returns that count. */
int64_t wpo_fft_process_kernel(int32_t length, const float *kernel, int32_t M,
                               float *out_spectrum)
{
    const int32_t N = wpo_next_pot(length + M);
    wpo_fft *f = wpo_fft_new(N);
    if (!f) return -1;
    int rc = padded_fft(f, kernel, M, out_spectrum);
    wpo_fft_free(f);
    return rc < 0 ? rc : 2 * (int64_t)N;
}

/* lib/convolve.js:564-586  fftConvolution(length, fftKernel, kernelLength)
 * -> go(input).  `fft` must have size nextPOT(length+kernelLength); out holds
 * length+kernelLength floats. */
int wpo_fft_convolution_go(const wpo_fft *f, int32_t length, const float *fft_kernel,
                           int32_t kernel_length, const float *input,
                           int64_t input_len, float *out)
{
    const int32_t result_length = length + kernel_length;
    float *spec = (float *)malloc(sizeof(float) * (size_t)f->csize);
    if (padded_fft(f, input, input_len, spec) < 0) {
        free(spec);
        return -1;
    }
    for (int32_t i = 0; i < f->csize; i += 2) {              /* :576-580 */
        const double ar = spec[i], ai = spec[i + 1];
        const double br = fft_kernel[i], bi = fft_kernel[i + 1];
        const double tmp = ar * br - ai * bi;
        spec[i + 1] = (float)(ar * bi + ai * br);
        spec[i] = (float)tmp;
    }
    float *conv = (float *)malloc(sizeof(float) * (size_t)f->csize);
    wpo_fft_inverse(f, conv, spec);                          /* :581 */
    for (int32_t i = 0; i < result_length; i++)              /* :582 */
        out[i] = (2 * (int64_t)i < f->csize) ? conv[2 * i] : 0.0f;
    free(conv);
    free(spec);
    return 0;
}

/* One-shot convenience: spectrum + go, as impulse-convolve.js:12,18 chain them. */
int wpo_convolve(const float *input, int64_t input_len, int32_t length,
                 const float *kernel, int32_t M, float *out)
{
    const int32_t N = wpo_next_pot(length + M);
    wpo_fft *f = wpo_fft_new(N);
    if (!f) return -1;
    float *spec = (float *)malloc(sizeof(float) * 2 * (size_t)N);
    int rc = padded_fft(f, kernel, M, spec);
    if (rc == 0) rc = wpo_fft_convolution_go(f, length, spec, M, input, input_len, out);
    free(spec);
    wpo_fft_free(f);
    return rc;
}

/* lib/convolve.js:541-551  directConvolution: f32 accumulator array,
 * each step: f64 product, f64 add, f32 store. out holds n+m-1 floats. */
void wpo_direct_convolution(const float *input, int64_t n, const float *kernel,
                            int64_t m, float *out)
{
    const int64_t len = n + m - 1;
    for (int64_t i = 0; i < len; i++) out[i] = 0.0f;
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < m; j++)
            out[i + j] = (float)((double)out[i + j] + (double)input[i] * (double)kernel[j]);
}

/* lib/impulse-reverb.js:14-16, impulse-reverb-stereo.js:18-21
 * curr[i] += prev[i]  (f64 add, f32 store) */
void wpo_add_tail(float *curr, const float *prev, int64_t n)
{
    for (int64_t i = 0; i < n; i++) curr[i] = (float)((double)curr[i] + (double)prev[i]);
}

/* ------------------------------------------------------------------------ */
/* src/mix-buffers.js:1-73.  One source folded into the target.
 *   sl  = (rl*length)|0 for the tuple form, rl for the plain form (:7,:13)
 *   vol = constant, or vol_vec[x] = _vol(x) evaluated by the caller (:21)
 *   o   = offset (:9)
 * Index rule (:22 etc.): (x+o) % sl % rl with JS remainder semantics: the
 * result takes the dividend's sign; a negative or NaN index reads `undefined`
 * which turns the sum into NaN; a -0 index names element 0.  sl==0 or rl==0
 * gives NaN.  sl<0 divides by |sl| (JS % ignores the divisor's sign). */
static inline double src_at(const float *s, int64_t x, int64_t o, int64_t sl, int64_t rl)
{
    if (sl == 0 || rl == 0) return NAN;
    const int64_t asl = sl < 0 ? -sl : sl;
    const int64_t i = ((x + o) % asl) % rl;
    if (i < 0) return NAN;
    return (double)s[i];
}

int wpo_mix_source(float *const *tgt, int tch, int64_t tl, const float *const *src,
                   int sch, int64_t rl, int64_t sl, double vol, const double *vol_vec,
                   int64_t o)
{
    if (tch == 2 && sch == 2) {                              /* :19-24, :48-52 */
        for (int64_t x = 0; x < tl; x++) {
            const double v = vol_vec ? vol_vec[x] : vol;
            tgt[0][x] = (float)((double)tgt[0][x] + src_at(src[0], x, o, sl, rl) * v);
            tgt[1][x] = (float)((double)tgt[1][x] + src_at(src[1], x, o, sl, rl) * v);
        }
    } else if (tch == 2 && sch == 1) {                       /* :25-30, :53-57 */
        for (int64_t x = 0; x < tl; x++) {
            const double v = vol_vec ? vol_vec[x] : vol;
            /* two reads, as in the reference: a source that shares memory with target[0] is read again after the first store */
            tgt[0][x] = (float)((double)tgt[0][x] + (src_at(src[0], x, o, sl, rl) / 2.0) * v);
            tgt[1][x] = (float)((double)tgt[1][x] + (src_at(src[0], x, o, sl, rl) / 2.0) * v);
        }
    } else if (tch == 1 && sch == 2) {                       /* :33-37, :60-63 */
        for (int64_t x = 0; x < tl; x++) {
            const double v = vol_vec ? vol_vec[x] : vol;
            const double a = src_at(src[0], x, o, sl, rl);
            const double b = src_at(src[1], x, o, sl, rl);
            tgt[0][x] = (float)((double)tgt[0][x] + ((a + b) / 2.0) * v);
        }
    } else if (tch == 1 && sch == 1) {                       /* :38-42, :64-67 */
        for (int64_t x = 0; x < tl; x++) {
            const double v = vol_vec ? vol_vec[x] : vol;
            tgt[0][x] = (float)((double)tgt[0][x] + src_at(src[0], x, o, sl, rl) * v);
        }
    }
    /* any other channel count: the reference silently does nothing */
    return 0;
}

/* src/mix-buffers.js:7  sl = (buffer[0].length * length)|0  (ToInt32 of a double) */
int32_t wpo_mix_sl(int64_t rl, double length)
{
    const double p = (double)rl * length;
    if (!isfinite(p)) return 0;
    double t = trunc(p);
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

/* ------------------------------------------------------------------------ */
/* lib/ring-buffer.js:10-22, 33-50: position a proxy access at `key` touches,
 * given the needle. Bit-exact integer rule. */
int64_t wpo_ring_pos(int64_t needle, int64_t key, int64_t length)
{
    int64_t pos = needle;
    if (key != 0) {
        pos += key;
        if (pos > length - 1) pos = pos % length;
        else if (pos < 0) while (pos < 0) pos += length;
    }
    return pos;
}

/* lib/ring-buffer.js:35-39: needle after a write at key 0 */
int64_t wpo_ring_advance(int64_t needle, int64_t length)
{
    needle++;
    if (needle > length - 1) needle = 0;
    return needle;
}

/* src/loop-node.js:145-147  beatRate = Math.round(sampleRate * (60 / bpm))
 * Math.round = floor(x + 0.5) for the positive values that occur here. */
int64_t wpo_beat_rate(double sample_rate, double bpm)
{
    return (int64_t)floor(sample_rate * (60.0 / bpm) + 0.5);
}

/* ------------------------------------------------------------------------ */
/* Synthetic inputs shared by every leg (reference run, oracle, HIP path).
 * Integer-hash noise (murmur3 finaliser) and a libm-free decaying IR so that
 * every platform produces the same bits. These define the test workload; they
 * restate nothing from the reference. */
static inline uint32_t fmix32(uint32_t h)
{
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}

void wpo_synth_noise(float *out, int64_t n, uint32_t seed, int64_t start)
{
    const uint32_t k = seed * 0x9E3779B9u;
    for (int64_t i = 0; i < n; i++) {
        const uint32_t u = fmix32((uint32_t)(start + i) ^ k);
        out[i] = (float)(((double)(u >> 8) - 8388608.0) / 8388608.0);
    }
}

void wpo_synth_ir(float *out, int64_t M, uint32_t seed)
{
    const uint32_t k = seed * 0x9E3779B9u;
    const float r = (float)(1.0 - 6.907755 / (double)M);
    float e = 1.0f;
    double norm = 0.0;
    for (int64_t i = 0; i < M; i++) {
        const uint32_t u = fmix32((uint32_t)i ^ k);
        const double nz = ((double)(u >> 8) - 8388608.0) / 8388608.0;
        out[i] = (float)(nz * (double)e);
        norm += fabs((double)out[i]);
        e = (float)((double)e * (double)r);
    }
    for (int64_t i = 0; i < M; i++) out[i] = (float)((double)out[i] / norm);
}

/* Order-sensitive 32-bit checksum of f32 bit patterns, h = h*31 + bits
 * (the same rule oracle/gen_golden.mjs applies to the reference's outputs). */
uint32_t wpo_checksum(const float *a, int64_t n)
{
    uint32_t h = 0;
    for (int64_t i = 0; i < n; i++) {
        uint32_t u;
        memcpy(&u, &a[i], 4);
        h = h * 31u + u;
    }
    return h;
}
