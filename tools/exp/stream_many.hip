// Read-only reduction over MANY streams, the master-reduce shape: S stereo stems of F frames
// each, every workgroup owns a tile of frames and walks all stems.  Variants: rows of 1024
// frames per tile (R), stems in flight (U), non-temporal loads, workgroup->tile order.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <int R, int U, bool NT, int ORDER>
__global__ __launch_bounds__(256) void k(const float *__restrict__ stems, float *__restrict__ bus, int64_t F, int S, int ntiles)
{
    // ORDER 0: tile = blockIdx.x; ORDER 1: tiles dealt so that the 8 XCDs (round-robin dispatch) get contiguous ranges
    int tile = blockIdx.x;
    if (ORDER == 1) {
        const int per = (ntiles + 7) / 8;
        tile = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (tile >= ntiles) return;
    }
    const int64_t xb = (int64_t)tile * (R * 1024) + threadIdx.x * 4;
    f4 a0[R], a1[R];
#pragma unroll
    for (int r = 0; r < R; r++) a0[r] = a1[r] = f4{0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 + U <= S; s0 += U) {
        f4 q0[U][R], q1[U][R];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const float *c0 = stems + (int64_t)(s0 + u) * 2 * F, *c1 = c0 + F;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const f4 *p0 = (const f4 *)(c0 + xb + r * 1024), *p1 = (const f4 *)(c1 + xb + r * 1024);
                q0[u][r] = NT ? __builtin_nontemporal_load(p0) : *p0;
                q1[u][r] = NT ? __builtin_nontemporal_load(p1) : *p1;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int r = 0; r < R; r++) { a0[r] += q0[u][r] * 0.5f; a1[r] += q1[u][r] * 0.5f; }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        *(f4 *)(bus + xb + r * 1024) = a0[r];
        *(f4 *)(bus + F + xb + r * 1024) = a1[r];
    }
}

template <int R, int U, bool NT, int ORDER>
static void run(const char *name, const float *stems, float *bus, int64_t F, int S)
{
    const int ntiles = (int)(F / (R * 1024));
    const int grid = ORDER == 1 ? ((ntiles + 7) / 8) * 8 : ntiles;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 3; it++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<R, U, NT, ORDER>), dim3(grid), dim3(256), 0, 0, stems, bus, F, S, ntiles);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    printf("%-40s %7.3f ms  %7.1f GB/s\n", name, best, (double)ntiles * R * 1024 * 8.0 * S / best / 1e6);
    fflush(stdout);
}

int main()
{
    const int64_t F = 2880000 / 4096 * 4096;  // whole tiles for every R
    const int S = 2048;
    float *stems, *bus;
    CHECK(hipMalloc((void **)&stems, (size_t)S * 2 * F * 4));
    CHECK(hipMalloc((void **)&bus, (size_t)2 * F * 4));
    CHECK(hipMemset(stems, 0, (size_t)S * 2 * F * 4));
    printf("S=%d stereo stems x F=%lld frames = %.1f GB\n", S, (long long)F, (double)S * 2 * F * 4 / 1e9);
    run<1, 8, false, 0>("R1 U8 plain (k_mix today)", stems, bus, F, S);
    run<1, 8, true, 0>("R1 U8 nt", stems, bus, F, S);
    run<1, 16, false, 0>("R1 U16 plain", stems, bus, F, S);
    run<4, 2, false, 0>("R4 U2 plain", stems, bus, F, S);
    run<4, 2, true, 0>("R4 U2 nt", stems, bus, F, S);
    run<4, 4, true, 0>("R4 U4 nt", stems, bus, F, S);
    run<2, 4, true, 0>("R2 U4 nt", stems, bus, F, S);
    run<1, 8, false, 1>("R1 U8 plain, XCD-contiguous tiles", stems, bus, F, S);
    run<1, 8, true, 1>("R1 U8 nt, XCD-contiguous tiles", stems, bus, F, S);
    run<4, 2, true, 1>("R4 U2 nt, XCD-contiguous tiles", stems, bus, F, S);
    return 0;
}
