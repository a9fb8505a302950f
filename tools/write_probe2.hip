// Diagnostic: is the HBM write rate a function of launch duration or of footprint?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(256) fill_region(d2* out, size_t region16, int passes)
{
    const d2 v = {1.0, 2.0};
    d2* base = out + (size_t)blockIdx.x * region16;
    for (int p = 0; p < passes; ++p)
        for (size_t i = threadIdx.x; i < region16; i += 256) __builtin_nontemporal_store(v, base + i);
}
int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 32;
    const size_t bytes = gib << 30, n16 = bytes / 16, region16 = 384 * 1024 / 16;
    d2* buf; CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fill_region, dim3((unsigned)(n16 / region16)), dim3(256), 0, 0, buf, region16, 1); CK(hipDeviceSynchronize());
    for (int split : {1, 2, 4, 8, 16}) for (int passes : {1, 4}) {
        const size_t sub16 = n16 / split, nblk = sub16 / region16;
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < 2; ++rep)
            for (int s = 0; s < split; ++s) hipLaunchKernelGGL(fill_region, dim3((unsigned)nblk), dim3(256), 0, 0, buf + s * sub16, region16, passes);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("buffer %zu GiB, %2d launches per sweep, %d passes per block: %8.1f GB/s  (%.2f ms per launch)\n", gib, split, passes,
               2.0 * passes * bytes / (ms * 1e-3) / 1e9, ms / (2 * split));
    }
    return 0;
}
