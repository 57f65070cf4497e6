// Does a working set that fits the 256 MiB Infinity Cache stream faster than HBM?
// Copies A -> B repeatedly for several working-set sizes (A + B), one 32 KB row per workgroup.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void k(const f4 *__restrict__ A, f4 *__restrict__ B)
{
    const int64_t row = blockIdx.x;
    const f4 *a = A + row * 2048;
    f4 *b = B + row * 2048;
    f4 v[8];
#pragma unroll
    for (int i = 0; i < 8; i++) v[i] = NT ? __builtin_nontemporal_load(a + threadIdx.x + 256 * i) : a[threadIdx.x + 256 * i];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        if (NT) __builtin_nontemporal_store(v[i] * 1.5f, b + threadIdx.x + 256 * i);
        else b[threadIdx.x + 256 * i] = v[i] * 1.5f;
    }
}

int main()
{
    void *A, *B;
    const size_t maxb = (size_t)4 << 30;
    CHECK(hipMalloc(&A, maxb));
    CHECK(hipMalloc(&B, maxb));
    CHECK(hipMemset(A, 0, maxb));
    CHECK(hipMemset(B, 0, maxb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int nt = 0; nt < 2; nt++)
        for (size_t mb : {16, 32, 64, 96, 128, 192, 256, 512, 1024, 4096}) {
            const int rows = (int)(mb * 1024 * 1024 / 32768);
            const int reps = (int)(8192 / mb) + 2;
            for (int w = 0; w < 2; w++) {
                if (nt) hipLaunchKernelGGL(k<true>, dim3(rows), dim3(256), 0, 0, (const f4 *)A, (f4 *)B);
                else hipLaunchKernelGGL(k<false>, dim3(rows), dim3(256), 0, 0, (const f4 *)A, (f4 *)B);
            }
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                if (nt) hipLaunchKernelGGL(k<true>, dim3(rows), dim3(256), 0, 0, (const f4 *)A, (f4 *)B);
                else hipLaunchKernelGGL(k<false>, dim3(rows), dim3(256), 0, 0, (const f4 *)A, (f4 *)B);
            }
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%s  A=%4zu MB + B=%4zu MB : %8.1f GB/s (%d launches, %.1f us each)\n", nt ? "nt   " : "plain", mb, mb,
                   2.0 * mb * 1048576.0 * reps / ms / 1e6, reps, ms * 1e3 / reps);
            fflush(stdout);
        }
    return 0;
}
