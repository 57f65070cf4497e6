// Read:write ratio sweep for a coalesced stream (32 KB rows, one workgroup per row):
// the ceilings that bound the FFT kernels (forward: reads 1, writes 2; inverse: reads 2, writes 1).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

// each workgroup handles one row of 4096 f2: reads R of every 2 elements' worth, writes Wn
template <int R, int Wn>
__global__ __launch_bounds__(256) void k(const f2 *__restrict__ A, f2 *__restrict__ Bo, int rows)
{
    const int64_t row = blockIdx.x;
    const f2 *a = A + row * 4096 * R / 2;
    f2 *b = Bo + row * 4096 * Wn / 2;
    f2 v[8 * R];
#pragma unroll
    for (int i = 0; i < 8 * R; i++) v[i] = __builtin_nontemporal_load(a + threadIdx.x + 256 * i);
    f2 s = {0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8 * R; i++) s += v[i];
#pragma unroll
    for (int i = 0; i < 8 * Wn; i++) __builtin_nontemporal_store(s + v[i % (8 * R)], b + threadIdx.x + 256 * i);
    if (Wn == 0 && s.x == 12345.678f) b[threadIdx.x] = s;  // read-only stream: keep the loads alive
}

// The same ratios with a software pipeline: every wave walks T rows, D loads in flight,
// one store per row (VI = floats read per lane per row, VO = floats written).
template <int VI, int VO, int D>
__global__ __launch_bounds__(256) void kp(const float *__restrict__ A, float *__restrict__ Bo, int T)
{
    typedef float vin __attribute__((ext_vector_type(VI)));
    typedef float vout __attribute__((ext_vector_type(VO)));
    const int64_t lane = (int64_t)blockIdx.x * 256 + threadIdx.x, width = (int64_t)gridDim.x * 256;
    const vin *a = (const vin *)A + lane;
    vout *b = (vout *)Bo + lane;
    vin ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = __builtin_nontemporal_load(a + (int64_t)d * width);
    for (int t = 0; t + D <= T; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            const vin v = ring[d];
            int tn = t + d + D;
            tn = tn < T ? tn : T - 1;
            ring[d] = __builtin_nontemporal_load(a + (int64_t)tn * width);
            vout o;
#pragma unroll
            for (int i = 0; i < VO; i++) o[i] = v[i % VI] * 1.5f;
            __builtin_nontemporal_store(o, b + (int64_t)(t + d) * width);
        }
    }
}

template <int VI, int VO, int D>
static void runp(void *A, void *B, size_t alloc_bytes)
{
    const int wgs = 8192, T = 336;  // 2 Mi lanes x 336 rows
    if ((size_t)wgs * 256 * T * 4 * (VI > VO ? VI : VO) > alloc_bytes) {  // checked on the host: never launch out of bounds
        printf("pipelined %d:%d skipped: buffers too small\n", VI, VO);
        return;
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((kp<VI, VO, D>), dim3(wgs), dim3(256), 0, 0, (const float *)A, (float *)B, T);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = (double)wgs * 256 * T * 4 * (VI + VO);
    printf("pipelined read:write %d:%d (ring %d)  %7.3f ms  %7.1f GB/s\n", VI, VO, D, best, bytes / best / 1e6);
    fflush(stdout);
}

template <int R, int Wn>
static void run(void *A, void *B, int rows)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<R, Wn>), dim3(rows), dim3(256), 0, 0, (const f2 *)A, (f2 *)B, rows);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = (double)rows * 4096 * 8 * (R + Wn) / 2;
    printf("read:write %d:%d  %7.3f ms  %7.1f GB/s\n", R, Wn, best, bytes / best / 1e6);
    fflush(stdout);
}

int main()
{
    const int rows = 360000;  // about the forward kernel's segment count
    const size_t bytes = (size_t)rows * 4096 * 8;
    void *A, *B;
    CHECK(hipMalloc(&A, bytes));
    CHECK(hipMalloc(&B, bytes));
    CHECK(hipMemset(A, 0, bytes));
    CHECK(hipMemset(B, 0, bytes));
    run<2, 2>(A, B, rows);
    run<1, 2>(A, B, rows);
    run<2, 1>(A, B, rows);
    run<2, 0>(A, B, rows);
    runp<2, 2, 6>(A, B, bytes);
    runp<2, 4, 6>(A, B, bytes);
    runp<1, 2, 6>(A, B, bytes);
    runp<4, 2, 6>(A, B, bytes);
    runp<2, 1, 6>(A, B, bytes);
    return 0;
}
