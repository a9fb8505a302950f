// Diagnostic (not part of the product): the sub-wave COOPERATIVE form of one Francis step that the review proposed,
// built and timed beside the product's single-lane step (ltp::francis_step_window4, ltp_roots.hpp) on the same 4 x 4 window.
// Cooperative form: four lanes per polynomial, lane c of a quad holds column c of the window (four doubles); the reflector set-up
// is computed redundantly by the four lanes from quad-broadcast elements (DPP quad_perm moves, two per double: gfx950's DPP on
// 64-bit operations only has row_newbcast); a LEFT reflector application is lane-local; a RIGHT application needs three columns'
// elements of every row in every lane (12 broadcasts) before each lane updates its own column. Every element sees the same
// floating-point operations in the same order as in the single-lane step: the probe checks that the two forms leave the SAME BITS.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-fast-math -std=c++17 -I longtermplanner_amd/csrc -o francis_coop_probe tools/francis_coop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "ltp_roots.hpp"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
using namespace ltp;

// the value lane K of the caller's quad holds
template <int K> __device__ __forceinline__ double qb(double x)
{
    const unsigned long long u = __builtin_bit_cast(unsigned long long, x);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, K * 0x55, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), K * 0x55, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

// one step on the window held as c[0..3] = T[0..3][j], j = lane & 3
__device__ __forceinline__ void step_coop(double (&c)[4], int j)
{
    const double T33 = qb<3>(c[3]), T22 = qb<2>(c[2]), T32 = qb<2>(c[3]), T23 = qb<3>(c[2]);
    const double sh0 = T33, sh1 = T22, sh2 = T32 * T23;
    const double T11 = qb<1>(c[1]), T21 = qb<1>(c[2]), T12 = qb<2>(c[1]), T10 = qb<0>(c[1]), T00 = qb<0>(c[0]), T01 = qb<1>(c[0]);
    int im;
    double v0, v1, v2;
    {
        const double r = sh0 - T11, s = sh1 - T11;
        v0 = (r * s - sh2) / T21 + T12;
        v1 = T22 - T11 - r - s;
        v2 = T32;
        im = 1;
        const double lhs = T10 * (rabs(v1) + rabs(v2));
        const double rhs = v0 * (rabs(T00) + rabs(T11) + rabs(T22));
        if (!(rabs(lhs) < kDblEps * rhs)) {
            const double r0 = sh0 - T00, s0 = sh1 - T00;
            v0 = (r0 * s0 - sh2) / T10 + T01;
            v1 = T11 - T00 - r0 - s0;
            v2 = T21;
            im = 0;
        }
    }
    if (im == 0) {
        double e0, e1, tau, beta;
        householder3(v0, v1, v2, e0, e1, tau, beta);
        if (beta != 0.0 && tau != 0.0) {
            {   // left: rows 0..2 of every column, lane-local
                double tmp = e0 * c[1] + e1 * c[2];
                tmp += c[0];
                c[0] -= tau * tmp; c[1] -= (tau * e0) * tmp; c[2] -= (tau * e1) * tmp;
            }
            const double g = j == 0 ? 1.0 : (j == 1 ? e0 : e1);
#pragma unroll
            for (int i = 0; i < 4; ++i) {   // right: columns 0..2 of every row
                double tmp = qb<1>(c[i]) * e0 + qb<2>(c[i]) * e1;
                tmp += qb<0>(c[i]);
                const double nv = c[i] - (tau * tmp) * g;          // (tau tmp) * 1.0 == tau tmp
                c[i] = j < 3 ? nv : c[i];
            }
        }
    }
    {
        const bool first = im == 1;
        double w0, w1, w2;
        if (first) { w0 = v0; w1 = v1; w2 = v2; } else { w0 = qb<0>(c[1]); w1 = qb<0>(c[2]); w2 = qb<0>(c[3]); }
        double e0, e1, tau, beta;
        householder3(w0, w1, w2, e0, e1, tau, beta);
        if (beta != 0.0) {
            if (j == 0) c[1] = first ? -c[1] : beta;
            if (tau != 0.0) {
                {   // left: rows 1..3 of columns 1..3
                    double tmp = e0 * c[2] + e1 * c[3];
                    tmp += c[1];
                    const double n1 = c[1] - tau * tmp, n2 = c[2] - (tau * e0) * tmp, n3 = c[3] - (tau * e1) * tmp;
                    if (j >= 1) { c[1] = n1; c[2] = n2; c[3] = n3; }
                }
                const double g = j == 1 ? 1.0 : (j == 2 ? e0 : e1);
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // right: columns 1..3 of every row
                    double tmp = qb<2>(c[i]) * e0 + qb<3>(c[i]) * e1;
                    tmp += qb<1>(c[i]);
                    const double nv = c[i] - (tau * tmp) * g;
                    c[i] = j >= 1 ? nv : c[i];
                }
            }
        }
    }
    {
        double e0, tau, beta;
        householder2(qb<1>(c[2]), qb<1>(c[3]), e0, tau, beta);
        if (beta != 0.0) {
            if (j == 1) c[2] = beta;
            if (tau != 0.0) {
                {   // left: rows 2, 3 of columns 2, 3
                    double tmp = e0 * c[3];
                    tmp += c[2];
                    const double n2 = c[2] - tau * tmp, n3 = c[3] - (tau * e0) * tmp;
                    if (j >= 2) { c[2] = n2; c[3] = n3; }
                }
                const double g = j == 2 ? 1.0 : e0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {   // right: columns 2, 3 of every row
                    double tmp = qb<3>(c[i]) * e0;
                    tmp += qb<2>(c[i]);
                    const double nv = c[i] - (tau * tmp) * g;
                    c[i] = j >= 2 ? nv : c[i];
                }
            }
        }
    }
    if (im == 0 && j == 0) { c[2] = 0.0; c[3] = 0.0; }
    if (j == 1) c[3] = 0.0;
}

// mode 0: single-lane step (lane 0, or all 64 lanes: the same time), mode 1: cooperative (quads)
__global__ void __launch_bounds__(64) probe(const double* t0, int steps, int mode, double* out, unsigned long long* ticks)
{
    const int lane = threadIdx.x, j = lane & 3;
    double T0[6][6], T[6][6];
    for (int i = 0; i < 6; ++i) for (int k = 0; k < 6; ++k) { T0[i][k] = t0[i * 6 + k]; T[i][k] = T0[i][k]; }
    double c0[4], c[4];
    for (int i = 0; i < 4; ++i) { c0[i] = t0[i * 6 + j]; c[i] = c0[i]; }
    __syncthreads();
    const unsigned long long a = wall_clock64();
    if (mode == 0) {
        for (int s = 0; s < steps; ++s) {
            francis_step_window4<6, double>(T, T[3][3], T[2][2], T[3][2] * T[2][3]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) T[i][k] = __builtin_fma(T[i][k], 1e-6, T0[i][k]);
        }
    } else {
        for (int s = 0; s < steps; ++s) {
            step_coop(c, j);
#pragma unroll
            for (int i = 0; i < 4; ++i) c[i] = __builtin_fma(c[i], 1e-6, c0[i]);
        }
    }
    const unsigned long long b = wall_clock64();
    if (mode == 0) { for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) if (lane == 0) out[i * 4 + k] = T[i][k]; }
    else { for (int i = 0; i < 4; ++i) if (lane < 4) out[i * 4 + j] = c[i]; }
    if (lane == 0) ticks[0] = b - a;
}

int main()
{
    double h[36] = {0};
    const double w[4][4] = {{0.31, -1.7, 0.42, 2.3}, {1.0, 0.12, -0.77, 0.5}, {0.0, 0.85, -0.21, 1.1}, {0.0, 0.0, 0.6, 0.44}};
    for (int i = 0; i < 4; ++i) for (int k = 0; k < 4; ++k) h[i * 6 + k] = w[i][k];
    double *dt, *dout; unsigned long long* dticks;
    CK(hipMalloc((void**)&dt, sizeof h)); CK(hipMalloc((void**)&dout, 16 * 8)); CK(hipMalloc((void**)&dticks, 8));
    CK(hipMemcpy(dt, h, sizeof h, hipMemcpyHostToDevice));
    double res[2][16];
    for (int steps : {1, 7, 4000}) {
        double ns[2];
        for (int mode = 0; mode < 2; ++mode) {
            unsigned long long best = ~0ull, t;
            for (int rep = 0; rep < 5; ++rep) {
                hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, nullptr, dt, steps, mode, dout, dticks);
                CK(hipDeviceSynchronize());
                CK(hipMemcpy(&t, dticks, 8, hipMemcpyDeviceToHost));
                best = t < best ? t : best;
            }
            CK(hipMemcpy(res[mode], dout, sizeof res[mode], hipMemcpyDeviceToHost));
            ns[mode] = best * 10.0 / steps;
        }
        const bool same = memcmp(res[0], res[1], sizeof res[0]) == 0;
        printf("%4d step(s): single-lane %8.1f ns per step, four lanes per polynomial %8.1f ns per step; the two windows afterwards: %s\n", steps, ns[0], ns[1],
               same ? "bit-identical" : "DIFFERENT");
    }
    return 0;
}
