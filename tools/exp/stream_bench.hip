// Memory-pattern microbenchmark for the spectral MAC's streaming shape: every wave
// walks T rows of a [jobs][T][B] complex-f32 array, reading its slice of each row and
// writing the same slice of a second array, with a ring of loads in flight.
// Prints achieved GB/s (read + write bytes) per variant.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <class V, bool NT> __device__ __forceinline__ V ld(const V *p) { return NT ? __builtin_nontemporal_load(p) : *p; }
template <class V, bool NT> __device__ __forceinline__ void st(V *p, V v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

// V: per-lane vector (f2 = one bin, f4 = two bins); D: ring depth; rowv: row length in V units.
// MODE 0: grid (rowv/WG, jobs): x = slice within the row (current MAC mapping)
// MODE 1: grid (jobs, rowv/WG): x = job (neighbouring workgroups on different jobs)
// MODE 2: layout [job][slice][t][WG lanes] (each workgroup's stream is contiguous)
template <class V, int D, bool NT, int MODE, int WG>
__global__ __launch_bounds__(WG) void k_stream(const V *__restrict__ Z, V *__restrict__ W, int T, int rowv)
{
    const int slice = MODE == 1 ? blockIdx.y : blockIdx.x;
    const int job = MODE == 1 ? blockIdx.x : blockIdx.y;
    const int64_t base = (int64_t)job * T * rowv;
    const int64_t step = MODE == 2 ? WG : rowv;
    if (MODE == 3 || MODE == 4) {
        // lanes 0-31 of a wave: bins q0..q0+31; lanes 32-63: the mirrored bins (row end), descending (3) or ascending (4)
        const int slot = slice * WG + threadIdx.x;
        const int q = ((slot >> 6) << 5) + (slot & 31);
        const int k = (slot & 32) ? (MODE == 3 ? rowv - 1 - q : rowv - 32 - (q & ~31) + (slot & 31)) : q;
        const V *z = Z + base + k;
        V *w = W + base + k;
        V ring[D];
#pragma unroll
        for (int d = 0; d < D; d++) ring[d] = ld<V, NT>(z + (int64_t)(d < T ? d : T - 1) * rowv);
        int t = 0;
        for (; t + D <= T; t += D) {
#pragma unroll
            for (int d = 0; d < D; d++) {
                V v = ring[d];
                int tn = t + d + D;
                tn = tn < T ? tn : T - 1;
                ring[d] = ld<V, NT>(z + (int64_t)tn * rowv);
                v = v * 1.0001f + 0.5f;
                st<V, NT>(w + (int64_t)(t + d) * rowv, v);
            }
        }
        for (int d = 0; t + d < T; d++) st<V, NT>(w + (int64_t)(t + d) * rowv, ring[d]);
        return;
    }
    const int64_t off = MODE == 2 ? base + (int64_t)slice * T * WG + threadIdx.x : base + (int64_t)slice * WG + threadIdx.x;
    const V *z = Z + off;
    V *w = W + off;
    V ring[D];
#pragma unroll
    for (int d = 0; d < D; d++) ring[d] = ld<V, NT>(z + (int64_t)(d < T ? d : T - 1) * step);
    int t = 0;
    for (; t + D <= T; t += D) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            V v = ring[d];
            int tn = t + d + D;
            tn = tn < T ? tn : T - 1;
            ring[d] = ld<V, NT>(z + (int64_t)tn * step);
            v = v * 1.0001f + 0.5f;
            st<V, NT>(w + (int64_t)(t + d) * step, v);
        }
    }
    for (int d = 0; t + d < T; d++) st<V, NT>(w + (int64_t)(t + d) * step, ring[d]);
}

template <class V, int D, bool NT, int MODE, int WG>
static void run(const char *name, void *Z, void *W, int jobs, int T, int B, int lds = 0)
{
    const int rowv = B * 8 / (int)sizeof(V);
    if (lds > 48 * 1024) CHECK(hipFuncSetAttribute((const void *)k_stream<V, D, NT, MODE, WG>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    dim3 grid = MODE == 1 ? dim3(jobs, rowv / WG) : dim3(rowv / WG, jobs);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 4; it++) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_stream<V, D, NT, MODE, WG>), grid, dim3(WG), lds, 0, (const V *)Z, (V *)W, T, rowv);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    const double bytes = 2.0 * jobs * (double)T * B * 8;
    printf("%-44s %7.3f ms  %7.1f GB/s\n", name, best, bytes / best / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int jobs = argc > 1 ? atoi(argv[1]) : 512, T = argc > 2 ? atoi(argv[2]) : 727, B = 4096;
    const size_t bytes = (size_t)jobs * T * B * 8;
    void *Z, *W;
    CHECK(hipMalloc(&Z, bytes));
    CHECK(hipMalloc(&W, bytes));
    CHECK(hipMemset(Z, 0, bytes));
    CHECK(hipMemset(W, 0, bytes));
    printf("jobs=%d T=%d B=%d  %.1f GB per array\n", jobs, T, B, bytes / 1e9);
    // device-to-device copy for reference
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            CHECK(hipEventRecord(e0));
            CHECK(hipMemcpyAsync(W, Z, bytes, hipMemcpyDeviceToDevice, 0));
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("%-44s %7.3f ms  %7.1f GB/s\n", "hipMemcpy D2D", best, 2.0 * bytes / best / 1e6);
    }
    run<f2, 6, true, 0, 256>("8B/lane D6 nt row-slices (MAC today)", Z, W, jobs, T, B);
    run<f2, 6, false, 0, 256>("8B/lane D6 plain", Z, W, jobs, T, B);
    run<f2, 12, true, 0, 256>("8B/lane D12 nt", Z, W, jobs, T, B);
    run<f2, 24, true, 0, 256>("8B/lane D24 nt", Z, W, jobs, T, B);
    run<f4, 6, true, 0, 256>("16B/lane D6 nt", Z, W, jobs, T, B);
    run<f4, 12, true, 0, 256>("16B/lane D12 nt", Z, W, jobs, T, B);
    run<f4, 6, false, 0, 256>("16B/lane D6 plain", Z, W, jobs, T, B);
    run<f2, 6, true, 1, 256>("8B/lane D6 nt job-major grid", Z, W, jobs, T, B);
    run<f2, 6, true, 2, 256>("8B/lane D6 nt contiguous per WG", Z, W, jobs, T, B);
    run<f4, 6, true, 2, 256>("16B/lane D6 nt contiguous per WG", Z, W, jobs, T, B);
    run<f2, 6, true, 0, 64>("8B/lane D6 nt WG=64", Z, W, jobs, T, B);
    run<f2, 6, true, 0, 512>("8B/lane D6 nt WG=512", Z, W, jobs, T, B);
    run<f2, 6, true, 0, 1024>("8B/lane D6 nt WG=1024", Z, W, jobs, T, B);
    run<f2, 6, true, 3, 256>("8B D6 nt mirrored halves, descending", Z, W, jobs, T, B);
    run<f2, 6, true, 4, 256>("8B D6 nt mirrored halves, ascending", Z, W, jobs, T, B);
    run<f2, 6, true, 3, 256>("3 waves/SIMD: mirrored descending", Z, W, jobs, T, B, 53 * 1024);
    run<f2, 6, true, 4, 256>("3 waves/SIMD: mirrored ascending", Z, W, jobs, T, B, 53 * 1024);
    // occupancy limited through LDS: 53 KB per 256-lane workgroup = 3 workgroups per CU = 3 waves per SIMD
    const int l3 = 53 * 1024, l2 = 80 * 1024, l4 = 40 * 1024;
    run<f2, 6, true, 0, 256>("3 waves/SIMD: 8B D6 nt row-slices", Z, W, jobs, T, B, l3);
    run<f2, 12, true, 0, 256>("3 waves/SIMD: 8B D12 nt row-slices", Z, W, jobs, T, B, l3);
    run<f2, 24, true, 0, 256>("3 waves/SIMD: 8B D24 nt row-slices", Z, W, jobs, T, B, l3);
    run<f2, 6, true, 1, 256>("3 waves/SIMD: 8B D6 nt job-major", Z, W, jobs, T, B, l3);
    run<f2, 12, true, 1, 256>("3 waves/SIMD: 8B D12 nt job-major", Z, W, jobs, T, B, l3);
    run<f4, 6, true, 0, 256>("3 waves/SIMD: 16B D6 nt row-slices", Z, W, jobs, T, B, l3);
    run<f4, 12, true, 0, 256>("3 waves/SIMD: 16B D12 nt row-slices", Z, W, jobs, T, B, l3);
    run<f2, 6, true, 0, 256>("2 waves/SIMD: 8B D6 nt row-slices", Z, W, jobs, T, B, l2);
    run<f2, 12, true, 0, 256>("2 waves/SIMD: 8B D12 nt row-slices", Z, W, jobs, T, B, l2);
    run<f4, 12, true, 0, 256>("2 waves/SIMD: 16B D12 nt row-slices", Z, W, jobs, T, B, l2);
    run<f2, 6, true, 0, 256>("4 waves/SIMD: 8B D6 nt row-slices", Z, W, jobs, T, B, l4);
    run<f2, 6, true, 1, 256>("4 waves/SIMD: 8B D6 nt job-major", Z, W, jobs, T, B, l4);
    return 0;
}
