// wp_aux_kernels.hip — the two parts of lib/convolve.js that are not on the block
// path but belong to its interface:
//   * directConvolution (:541-551), reproduced bit for bit: every output sample is
//     the reference's own chain of f64 multiply, f64 add, f32 store, in the
//     reference's order of the input index;
//   * the reference-layout kernel object: the interleaved N-point spectrum that
//     fftProcessKernel (:553-562) returns, and the way back from such a spectrum to
//     the impulse response, so that kernel objects produced by the reference itself
//     are accepted.  A radix-2 Stockham transform with binary64 butterflies over
//     global memory: it runs once per impulse response, not per block.
// Built with -ffp-contract=off (separately rounded multiply and add are part of the
// direct sum's contract).
#include <hip/hip_runtime.h>
#include <math.h>

#include "wp_internal.h"

namespace wp {

// out[k] = fold over i ascending of  f32( f64(out[k]) + f64(x[i]) * f64(h[k-i]) ), out[k] starting at 0:
// the value the reference's doubly nested loop leaves in its Float32Array accumulator
__global__ __launch_bounds__(256) void k_direct_conv(const float *__restrict__ x, int64_t n, const float *__restrict__ h,
                                                     int64_t m, float *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= n + m - 1) return;
    const int64_t lo = k - (m - 1) > 0 ? k - (m - 1) : 0;
    const int64_t hi = k < n - 1 ? k : n - 1;
    float acc = 0.0f;
    for (int64_t i = lo; i <= hi; i++) acc = (float)((double)acc + (double)x[i] * (double)h[k - i]);
    out[k] = acc;
}

hipError_t launch_direct_conv(const float *x, int64_t n, const float *h, int64_t m, float *out, hipStream_t s)
{
    const int64_t len = n + m - 1;
    if (len <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_direct_conv, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, x, n, h, m, out);
    return hipGetLastError();
}

struct c64 {
    double x, y;
};

// real samples (zero-padded to N) or an interleaved f32 spectrum -> binary64 complex
__global__ void k_big_load(const float *__restrict__ src, int64_t n_real, int interleaved, c64 *__restrict__ dst, int64_t N)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    c64 v;
    if (interleaved) {
        v.x = src[2 * i];
        v.y = src[2 * i + 1];
    } else {
        v.x = i < n_real ? (double)src[i] : 0.0;
        v.y = 0.0;
    }
    dst[i] = v;
}

// one autosort radix-2 stage: butterflies j in [0, N/2), Ns = product of the earlier radices
__global__ void k_big_stage(const c64 *__restrict__ in, c64 *__restrict__ out, int64_t N, int64_t Ns, int inverse)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N / 2) return;
    const int64_t k = j & (Ns - 1);
    double sn, cs;
    sincospi((double)k / (double)Ns, &sn, &cs);  // exp(-+ i*pi*k/Ns)
    if (!inverse) sn = -sn;
    const c64 a = in[j], b = in[j + N / 2];
    const c64 t = {b.x * cs - b.y * sn, b.x * sn + b.y * cs};
    const int64_t o = ((j - k) << 1) + k;
    out[o] = c64{a.x + t.x, a.y + t.y};
    out[o + Ns] = c64{a.x - t.x, a.y - t.y};
}

// forward: interleaved f32 spectrum (2N floats); inverse: real parts / N, first n_out samples
__global__ void k_big_store(const c64 *__restrict__ src, int64_t N, int inverse, float *__restrict__ dst, int64_t n_out)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (inverse) {
        if (i < n_out) dst[i] = (float)(src[i].x / (double)N);
    } else if (i < N) {
        dst[2 * i] = (float)src[i].x;
        dst[2 * i + 1] = (float)src[i].y;
    }
}

// src: n_real real samples (forward) or 2N interleaved floats (inverse); work0/work1: N binary64 complex each
hipError_t launch_big_fft(const float *src, int64_t n_real, int64_t N, int inverse, void *work0, void *work1, float *dst,
                          int64_t n_out, hipStream_t s)
{
    c64 *a = reinterpret_cast<c64 *>(work0), *b = reinterpret_cast<c64 *>(work1);
    const dim3 block(256), gridN((unsigned)((N + 255) / 256)), gridH((unsigned)((N / 2 + 255) / 256));
    hipLaunchKernelGGL(k_big_load, gridN, block, 0, s, src, n_real, inverse, a, N);
    for (int64_t Ns = 1; Ns < N; Ns <<= 1) {
        hipLaunchKernelGGL(k_big_stage, gridH, block, 0, s, a, b, N, Ns, inverse);
        c64 *t = a;
        a = b;
        b = t;
    }
    hipLaunchKernelGGL(k_big_store, gridN, block, 0, s, a, N, inverse, dst, n_out);
    return hipGetLastError();
}

}  // namespace wp
