// Diagnostic (not part of the product; VERDICT r5 item 2, "first measure one Francis step on one lane"): where the time of ONE step of the
// device eigen-solve goes when a lone lane of a lone wave runs it — the situation of the slowest polynomial of queue B, which decides
// k_scaling_slow. The step is ltp::francis_step_window4 (ltp_roots.hpp: the 4 x 4 window on which 92 % of all degree-6 steps are taken);
// the variants below are copies of it with parts taken out:
//   full          the product's step (three reflectors: set-up = 1 sqrt + 3 divisions each, then row and column updates)
//   no_setup      the reflector set-up replaced by three multiplications (no square root, no division); updates as in the product
//   no_updates    the set-up as in the product; the row / column updates left out (one element touched so that nothing is dead code)
//   shifts_only   neither: the shift computation, the start-row test (two divisions) and the loop
// Every variant runs K steps from a matrix that is pulled back towards its start after every step (16 fused multiply-adds, the same in
// all variants), on lane 0 of one wave, timed with the wall clock (100 MHz).
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -I longtermplanner_amd/csrc -o francis_split_probe tools/francis_split_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "ltp_roots.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace ltp;

template <bool SETUP> __device__ __forceinline__ void hh3(double w0, double w1, double w2, double& e0, double& e1, double& tau, double& beta)
{
    if constexpr (SETUP) householder3(w0, w1, w2, e0, e1, tau, beta);
    else { e0 = w1 * 0.25; e1 = w2 * 0.25; tau = 1.25; beta = w0 * 1.5; }
}
template <bool SETUP> __device__ __forceinline__ void hh2(double w0, double w1, double& e0, double& tau, double& beta)
{
    if constexpr (SETUP) householder2(w0, w1, e0, tau, beta);
    else { e0 = w1 * 0.25; tau = 1.25; beta = w0 * 1.5; }
}

// francis_step_window4 of ltp_roots.hpp with SETUP / UPDATES switchable (N = 6 storage, window rows / columns 0..3)
template <bool SETUP, bool UPDATES>
__device__ __forceinline__ void step_variant(double (&T)[6][6], double sh0, double sh1, double sh2)
{
    int im;
    double v0, v1, v2;
    {
        const double Tmm = T[1][1];
        const double r = sh0 - Tmm, s = sh1 - Tmm;
        v0 = (r * s - sh2) / T[2][1] + T[1][2];
        v1 = T[2][2] - Tmm - r - s;
        v2 = T[3][2];
        im = 1;
        const double lhs = T[1][0] * (rabs(v1) + rabs(v2));
        const double rhs = v0 * (rabs(T[0][0]) + rabs(Tmm) + rabs(T[2][2]));
        if (!(rabs(lhs) < kDblEps * rhs)) {
            const double T00 = T[0][0];
            const double r0 = sh0 - T00, s0 = sh1 - T00;
            v0 = (r0 * s0 - sh2) / T[1][0] + T[0][1];
            v1 = T[1][1] - T00 - r0 - s0;
            v2 = T[2][1];
            im = 0;
        }
    }
    if (im == 0) {
        double e0, e1, tau, beta;
        hh3<SETUP>(v0, v1, v2, e0, e1, tau, beta);
        if (beta != 0.0 && tau != 0.0) {
            if constexpr (UPDATES) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    double tmp = e0 * T[1][j] + e1 * T[2][j];
                    tmp += T[0][j];
                    T[0][j] -= tau * tmp; T[1][j] -= (tau * e0) * tmp; T[2][j] -= (tau * e1) * tmp;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double tmp = T[i][1] * e0 + T[i][2] * e1;
                    tmp += T[i][0];
                    T[i][0] -= tau * tmp; T[i][1] -= (tau * tmp) * e0; T[i][2] -= (tau * tmp) * e1;
                }
            } else { T[1][0] += (e0 + e1) * tau; T[2][0] += beta; }
        }
    }
    {
        const bool first = im == 1;
        double w0, w1, w2;
        if (first) { w0 = v0; w1 = v1; w2 = v2; } else { w0 = T[1][0]; w1 = T[2][0]; w2 = T[3][0]; }
        double e0, e1, tau, beta;
        hh3<SETUP>(w0, w1, w2, e0, e1, tau, beta);
        if (beta != 0.0) {
            if (first) T[1][0] = -T[1][0]; else T[1][0] = beta;
            if (tau != 0.0) {
                if constexpr (UPDATES) {
#pragma unroll
                    for (int j = 1; j < 4; ++j) {
                        double tmp = e0 * T[2][j] + e1 * T[3][j];
                        tmp += T[1][j];
                        T[1][j] -= tau * tmp; T[2][j] -= (tau * e0) * tmp; T[3][j] -= (tau * e1) * tmp;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double tmp = T[i][2] * e0 + T[i][3] * e1;
                        tmp += T[i][1];
                        T[i][1] -= tau * tmp; T[i][2] -= (tau * tmp) * e0; T[i][3] -= (tau * tmp) * e1;
                    }
                } else { T[2][1] += (e0 + e1) * tau; }
            }
        }
    }
    {
        double e0, tau, beta;
        hh2<SETUP>(T[2][1], T[3][1], e0, tau, beta);
        if (beta != 0.0) {
            T[2][1] = beta;
            if (tau != 0.0) {
                if constexpr (UPDATES) {
#pragma unroll
                    for (int j = 2; j < 4; ++j) {
                        double tmp = e0 * T[3][j];
                        tmp += T[2][j];
                        T[2][j] -= tau * tmp; T[3][j] -= (tau * e0) * tmp;
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double tmp = T[i][3] * e0;
                        tmp += T[i][2];
                        T[i][2] -= tau * tmp; T[i][3] -= (tau * tmp) * e0;
                    }
                } else { T[3][2] += e0 * tau; }
            }
        }
    }
    if (im == 0) { T[2][0] = 0.0; T[3][0] = 0.0; }
    T[3][1] = 0.0;
}


// a TUNED single-lane step (same operations on the same elements): the two possible start-row quotients are formed side by side
// (their division chains overlap), and the "reflector is degenerate" guards are decided for the whole wave (one ballot, a scalar
// branch) instead of as lane masks around the updates
__device__ __forceinline__ void step_tuned(double (&T)[6][6], double sh0, double sh1, double sh2)
{
    int im;
    double v0, v1, v2;
    {
        const double Tmm = T[1][1], T00 = T[0][0];
        const double r = sh0 - Tmm, s = sh1 - Tmm, r0 = sh0 - T00, s0 = sh1 - T00;
        const double qa = (r * s - sh2) / T[2][1], qb_ = (r0 * s0 - sh2) / T[1][0];
        v0 = qa + T[1][2];
        v1 = T[2][2] - Tmm - r - s;
        v2 = T[3][2];
        const double lhs = T[1][0] * (rabs(v1) + rabs(v2));
        const double rhs = v0 * (rabs(T00) + rabs(Tmm) + rabs(T[2][2]));
        const bool start1 = rabs(lhs) < kDblEps * rhs;
        im = start1 ? 1 : 0;
        v0 = start1 ? v0 : qb_ + T[0][1];
        v1 = start1 ? v1 : T[1][1] - T00 - r0 - s0;
        v2 = start1 ? v2 : T[2][1];
    }
    const bool uniform0 = __builtin_amdgcn_ballot_w64(im != 0) == 0ull;
    if (im == 0) {
        double e0, e1, tau, beta;
        householder3(v0, v1, v2, e0, e1, tau, beta);
        const bool ok = beta != 0.0 && tau != 0.0;
        if (uniform0 && __builtin_amdgcn_ballot_w64(!ok) == 0ull) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double tmp = e0 * T[1][j] + e1 * T[2][j];
                tmp += T[0][j];
                T[0][j] -= tau * tmp; T[1][j] -= (tau * e0) * tmp; T[2][j] -= (tau * e1) * tmp;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double tmp = T[i][1] * e0 + T[i][2] * e1;
                tmp += T[i][0];
                T[i][0] -= tau * tmp; T[i][1] -= (tau * tmp) * e0; T[i][2] -= (tau * tmp) * e1;
            }
        } else if (ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double tmp = e0 * T[1][j] + e1 * T[2][j];
                tmp += T[0][j];
                T[0][j] -= tau * tmp; T[1][j] -= (tau * e0) * tmp; T[2][j] -= (tau * e1) * tmp;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                double tmp = T[i][1] * e0 + T[i][2] * e1;
                tmp += T[i][0];
                T[i][0] -= tau * tmp; T[i][1] -= (tau * tmp) * e0; T[i][2] -= (tau * tmp) * e1;
            }
        }
    }
    {
        const bool first = im == 1;
        double w0, w1, w2;
        if (first) { w0 = v0; w1 = v1; w2 = v2; } else { w0 = T[1][0]; w1 = T[2][0]; w2 = T[3][0]; }
        double e0, e1, tau, beta;
        householder3(w0, w1, w2, e0, e1, tau, beta);
        if (beta != 0.0) {
            if (first) T[1][0] = -T[1][0]; else T[1][0] = beta;
            if (tau != 0.0) {
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    double tmp = e0 * T[2][j] + e1 * T[3][j];
                    tmp += T[1][j];
                    T[1][j] -= tau * tmp; T[2][j] -= (tau * e0) * tmp; T[3][j] -= (tau * e1) * tmp;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double tmp = T[i][2] * e0 + T[i][3] * e1;
                    tmp += T[i][1];
                    T[i][1] -= tau * tmp; T[i][2] -= (tau * tmp) * e0; T[i][3] -= (tau * tmp) * e1;
                }
            }
        }
    }
    {
        double e0, tau, beta;
        householder2(T[2][1], T[3][1], e0, tau, beta);
        if (beta != 0.0) {
            T[2][1] = beta;
            if (tau != 0.0) {
#pragma unroll
                for (int j = 2; j < 4; ++j) {
                    double tmp = e0 * T[3][j];
                    tmp += T[2][j];
                    T[2][j] -= tau * tmp; T[3][j] -= (tau * e0) * tmp;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double tmp = T[i][3] * e0;
                    tmp += T[i][2];
                    T[i][2] -= tau * tmp; T[i][3] -= (tau * tmp) * e0;
                }
            }
        }
    }
    if (im == 0) { T[2][0] = 0.0; T[3][0] = 0.0; }
    T[3][1] = 0.0;
}

// mode 5: step_tuned. mode 0: the product's step; 1: no_setup; 2: no_updates; 3: shifts_only; 4: the pull-back loop alone
__global__ void __launch_bounds__(64) probe(const double* t0, int steps, int mode, int lanes, double* out, unsigned long long* ticks)
{
    double T0[6][6], T[6][6];
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) { T0[i][j] = t0[i * 6 + j] * (1.0 + 1e-3 * (threadIdx.x & 7)); T[i][j] = T0[i][j]; }
    __syncthreads();
    const unsigned long long c0 = wall_clock64();
    if ((int)threadIdx.x < lanes) {
        for (int s = 0; s < steps; ++s) {
            const double sh0 = T[3][3], sh1 = T[2][2], sh2 = T[3][2] * T[2][3];
            if (mode == 0) francis_step_window4<6, double>(T, sh0, sh1, sh2);
            else if (mode == 1) step_variant<false, true>(T, sh0, sh1, sh2);
            else if (mode == 2) step_variant<true, false>(T, sh0, sh1, sh2);
            else if (mode == 3) step_variant<false, false>(T, sh0, sh1, sh2);
            else if (mode == 5) step_tuned(T, sh0, sh1, sh2);
            // pull the window back towards its start: bounded values, the same 16 operations in every variant
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) T[i][j] = __builtin_fma(T[i][j], 1e-6, T0[i][j]);
        }
    }
    const unsigned long long c1 = wall_clock64();
    double acc = 0.0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc += T[i][j];
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) ticks[0] = c1 - c0;
}

int main()
{
    // an unreduced upper Hessenberg 4 x 4 window with a complex pair and two real eigenvalues (values of planner size)
    double h[36] = {0};
    const double w[4][4] = {{0.31, -1.7, 0.42, 2.3}, {1.0, 0.12, -0.77, 0.5}, {0.0, 0.85, -0.21, 1.1}, {0.0, 0.0, 0.6, 0.44}};
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) h[i * 6 + j] = w[i][j];
    double *dt, *dout; unsigned long long* dticks;
    CK(hipMalloc((void**)&dt, sizeof h)); CK(hipMalloc((void**)&dout, 64 * 8)); CK(hipMalloc((void**)&dticks, 8));
    CK(hipMemcpy(dt, h, sizeof h, hipMemcpyHostToDevice));
    const char* names[5] = {"full (francis_step_window4)", "no_setup (no sqrt / division in the reflectors)", "no_updates (set-up only)", "shifts_only", "pull-back loop alone"};
    const int steps = 4000;
    double ns[5][2];
    for (int lanes : {1, 64}) {
        for (int mode = 0; mode < 5; ++mode) {
            unsigned long long best = ~0ull, t;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, nullptr, dt, steps, mode, lanes, dout, dticks);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&t, dticks, 8, hipMemcpyDeviceToHost));
                best = t < best ? t : best;
            }
            ns[mode][lanes == 64] = best * 10.0 / steps;
        }
    }
    for (int lanes : {1, 64}) {
        unsigned long long best = ~0ull, t;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, nullptr, dt, steps, 5, lanes, dout, dticks);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(&t, dticks, 8, hipMemcpyDeviceToHost));
            best = t < best ? t : best;
        }
        printf("tuned single-lane step (both start-row quotients at once, wave-level guards), %d lane(s): %.1f ns per step\n", lanes, best * 10.0 / steps);
    }
    for (int col = 0; col < 2; ++col) {
        printf("%s active lane(s) of one wave, %d steps:\n", col ? "64" : "1", steps);
        for (int mode = 0; mode < 5; ++mode)
            printf("  %-52s %8.1f ns per step   (minus the loop: %7.1f)\n", names[mode], ns[mode][col], ns[mode][col] - ns[4][col]);
        const double full = ns[0][col] - ns[4][col], nos = ns[1][col] - ns[4][col], nou = ns[2][col] - ns[4][col], sh = ns[3][col] - ns[4][col];
        printf("  => reflector set-up (3 sqrt + 8 divisions): %.0f ns = %.0f %%; row / column updates: %.0f ns = %.0f %%; shifts + start-row test: %.0f ns = %.0f %%\n",
               full - nos, 100.0 * (full - nos) / full, full - nou, 100.0 * (full - nou) / full, sh, 100.0 * sh / full);
    }
    return 0;
}
