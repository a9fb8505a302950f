// Diagnostic (not part of the product): HBM write ceilings on MI355X for a few store patterns.
// build: hipcc --offload-arch=gfx950 -O3 -o write_probe write_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// P0: whole grid sweeps the buffer as one front (grid-stride)
__global__ void __launch_bounds__(256) fill_global(d2* out, size_t n16, int nt)
{
    const d2 v = {1.0, 2.0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) {
        if (nt) __builtin_nontemporal_store(v, out + i); else out[i] = v;
    }
}
// P1: each block owns a contiguous region of `region16` 16-B slots and writes it front to back in `streams` interleaved streams
__global__ void __launch_bounds__(256) fill_region(d2* out, size_t region16, int streams, int nt)
{
    const d2 v = {1.0, 2.0};
    d2* base = out + (size_t)blockIdx.x * region16;
    const size_t per = region16 / streams;
    for (size_t i = threadIdx.x; i < per; i += 256)
        for (int s = 0; s < streams; ++s) {
            if (nt) __builtin_nontemporal_store(v, base + s * per + i); else base[s * per + i] = v;
        }
}
int main(int argc, char** argv)
{
    const size_t gib = argc > 1 ? atoi(argv[1]) : 32;
    const size_t bytes = gib << 30, n16 = bytes / 16;
    d2* buf; CK(hipMalloc(&buf, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time = [&](const char* name, auto launch) {
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-44s %8.1f GB/s\n", name, 3.0 * bytes / (ms * 1e-3) / 1e9);
    };
    for (int nt = 0; nt < 2; ++nt) {
        char nm[128];
        for (int blocks : {2048, 8192}) {
            snprintf(nm, sizeof nm, "global front, %d blocks, nt=%d", blocks, nt);
            time(nm, [&] { hipLaunchKernelGGL(fill_global, dim3(blocks), dim3(256), 0, 0, buf, n16, nt); });
        }
        for (size_t region_kb : {64, 384, 4096}) for (int streams : {1, 4, 28}) {
            const size_t region16 = region_kb * 1024 / 16;
            if (region16 % streams) continue;
            const size_t nblk = n16 / region16;
            snprintf(nm, sizeof nm, "block regions %zu KB, %d streams, nt=%d", region_kb, streams, nt);
            time(nm, [&] { hipLaunchKernelGGL(fill_region, dim3((unsigned)nblk), dim3(256), 0, 0, buf, region16, streams, nt); });
        }
    }
    return 0;
}
