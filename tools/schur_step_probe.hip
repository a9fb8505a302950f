// Diagnostic (not part of the product): time of the device eigen-solve (smallest_positive_real_root<6>) for the 64 slowest and 64 ordinary
// degree-6 polynomials of a 100 k panda batch, one wave: lane 0 alone, its polynomial in all lanes, one polynomial per lane.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -I longtermplanner_amd/csrc -o tools/schur_step_probe tools/schur_step_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "ltp_roots.hpp"
#include "schur_step_probe_polys.inc"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ void __launch_bounds__(64) solve(const double* coef, double* root, unsigned long long* ticks, int mode /* 0: lane 0 only, 1: all lanes lane 0's polynomial, 2: lane i polynomial i, 3: lane 0 the slowest polynomial, the others ordinary ones */)
{
    const int lane = threadIdx.x;
    const double* c = coef + ((mode == 2 || (mode == 3 && lane > 0)) ? lane + (mode == 3 ? 64 : 0) : 0) * 7;
    double p[7];
    for (int i = 0; i < 7; ++i) p[i] = c[i];
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    double r = 0.0;
    if (mode != 0 || lane == 0) r = ltp::smallest_positive_real_root<6>(p);
    const unsigned long long t1 = wall_clock64();
    root[lane] = r;
    if (lane == 0) ticks[0] = t1 - t0;
}
// mode 3's wave as wave 7 of a block of eight (k_scaling_slow's shape: eight candidate waves, two per SIMD), the others solving ordinary
// polynomials, in `blocks` blocks: does the slowest wave lose time to its neighbours?
__global__ void __launch_bounds__(512) solve8(const double* coef, double* root, unsigned long long* ticks)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double* c = coef + ((wave == 7 && lane == 0) ? 0 : (64 + lane)) * 7;
    double p[7];
    for (int i = 0; i < 7; ++i) p[i] = c[i];
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    const double r = ltp::smallest_positive_real_root<6>(p);
    const unsigned long long t1 = wall_clock64();
    if (blockIdx.x == 0) root[threadIdx.x & 63] = r;
    if (wave == 7 && lane == 0) atomicMax(&ticks[0], t1 - t0);
    __syncthreads();
}
int main()
{
    double *dc, *dr; unsigned long long* dt;
    CK(hipMalloc((void**)&dc, 2 * sizeof(kWorst))); CK(hipMalloc((void**)&dr, 64 * 8)); CK(hipMalloc((void**)&dt, 8));
    for (int set = 0; set < 2; ++set) {
        CK(hipMemcpy(dc, set == 0 ? kWorst : kTypical, sizeof(kWorst), hipMemcpyHostToDevice));
        CK(hipMemcpy(dc + 64 * 7, kTypical, sizeof(kTypical), hipMemcpyHostToDevice));
        for (int mode = 0; mode < (set == 0 ? 4 : 3); ++mode) {
            unsigned long long best = ~0ull, h;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(solve, dim3(1), dim3(64), 0, nullptr, dc, dr, dt, mode);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&h, dt, 8, hipMemcpyDeviceToHost));
                best = h < best ? h : best;
            }
            double r0; CK(hipMemcpy(&r0, dr, 8, hipMemcpyDeviceToHost));
            printf("%s polynomials, mode %d: %.2f us (lane 0: %d steps -> %.0f ns per step if it were alone); root %.17g\n", set == 0 ? "worst" : "typical", mode, best * 0.01,
                   set == 0 ? kWorstSteps[0] : 0, set == 0 ? best * 10.0 / kWorstSteps[0] : 0.0, r0);
        }
    }
    CK(hipMemcpy(dc, kWorst, sizeof(kWorst), hipMemcpyHostToDevice));
    CK(hipMemcpy(dc + 64 * 7, kTypical, sizeof(kTypical), hipMemcpyHostToDevice));
    for (int blocks : {1, 20, 256}) {
        unsigned long long best = ~0ull, h, z = 0;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemcpy(dt, &z, 8, hipMemcpyHostToDevice));
            hipLaunchKernelGGL(solve8, dim3(blocks), dim3(512), 0, nullptr, dc, dr, dt);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&h, dt, 8, hipMemcpyDeviceToHost));
            best = h < best ? h : best;
        }
        printf("the slowest polynomial in wave 7 of eight-wave blocks, %d block(s): %.2f us\n", blocks, best * 0.01);
    }
    return 0;
}
