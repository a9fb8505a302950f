// ltp_aux_kernels.hip — synthetic query generator, the one-lane entry points behind the reference's protected members and
// the parity probes, gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile). No fast-math:
// the inf/NaN flow of the reference is part of the contract.
#include "ltp_device.hpp"

namespace ltp {

// ---------------------------------------------------------------------------------------
// Synthetic queries (distribution of reference tests/randomConfiguration.m:14-34
// generalised to per-joint limits). Counter-based: value = f(seed, query, joint, field), so any
// shard of any batch can be generated independently and the host reproduces it bit for bit.
// ---------------------------------------------------------------------------------------
LTP_DEV double unit_random(unsigned long long seed, unsigned long long query, unsigned int joint, unsigned int field)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((query * 64ull + joint) * 4ull + field) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * 0x1.0p-53;
}

__global__ void __launch_bounds__(256)
k_generate(long long n, int dof, Limits lim, unsigned long long seed, long long first_query,
           double* __restrict__ q_goal, double* __restrict__ q_0, double* __restrict__ v_0, double* __restrict__ a_0,
           long long sq, long long sj)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * dof) return;
    const long long q = idx / dof;
    const int j = (int)(idx - q * dof);
    const JointLimits L = load_limits(lim, j);
    const unsigned long long gq = (unsigned long long)(first_query + q);
    const double eps = 1e-6;
    const double u0 = unit_random(seed, gq, j, 0), u1 = unit_random(seed, gq, j, 1);
    const double u2 = unit_random(seed, gq, j, 2), u3 = unit_random(seed, gq, j, 3);
    const double q0 = L.q_min + u0 * (L.q_max - L.q_min);
    const double qg = L.q_min + u1 * (L.q_max - L.q_min);
    const double vm = L.v_max - eps;
    const double v0 = -vm + u2 * (2.0 * vm);
    double a_lb, a_ub;
    if (v0 >= 0.0) {
        a_lb = -(L.a_max - eps);
        a_ub = dmin(L.a_max - eps, dsqrt(2.0 * L.j_max * (L.v_max - v0)));
    } else {
        a_lb = dmax(-(L.a_max - eps), -dsqrt(2.0 * L.j_max * (L.v_max - dabs(v0))));
        a_ub = L.a_max;
    }
    const double a0 = a_lb + u3 * (a_ub - a_lb);
    const long long ix = q * sq + (long long)j * sj;
    q_goal[ix] = qg;
    q_0[ix] = q0;
    v_0[ix] = v0;
    a_0[ix] = a0;
}

// ---------------------------------------------------------------------------------------
// One-lane mirrors of the protected member functions (for the reference's KAT-style tests).
// ---------------------------------------------------------------------------------------
// LimPow of every joint under one pow rule (ltp_profile.hpp): the same device functions, on the same bits, that the kernels would
// call — formed once per ltp_set_limits instead of once per lane. One thread per joint; out[j * kLimPowN + ...].
template <int SEM>
__global__ void __launch_bounds__(64) k_limit_powers(int dof, const double* a_max, const double* j_max, double* out)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();
    for (int j = threadIdx.x; j < dof; j += blockDim.x) {
        const double am = a_max[j], jm = j_max[j];
        const double tj = am / jm;                     // cc:124, 171, 186: t_rel = a_max / j_max
        double* w = out + (long long)j * kLimPowN;
        w[0] = tj;
        w[1] = pw3<SEM>(tj);
        w[2] = pw4<SEM>(tj);
        w[3] = pw3<SEM>(am);
        w[4] = pw4<SEM>(am);
        w[5] = pw3<SEM>(jm);
        w[6] = pw4<SEM>(jm);
    }
}

// LongTermPlanner::checkInputs (cc:68-77) for one query
template <int SEM>
__global__ void k_check_inputs(int dof, Limits lim, const double* q_0, const double* v_0, const double* a_0, int* ok)
{
    int good = 1;
    for (int j = 0; j < dof; ++j)
        if (!check_inputs_joint<SEM>(load_limits(lim, j), q_0[j], v_0[j], a_0[j])) good = 0;
    *ok = good;
}

template <int SEM>
__global__ void k_single_opt_braking(int joint, double t_sample, Limits lim, double v_0, double a_0, double* out)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();
    const JointLimits L = load_limits(lim, joint);
    double r[7] = {out[0], out[1], out[2], out[3], out[4], out[5], out[6]};
    double q, dir;
    MatlabCtx mc;
    opt_braking<SEM>(L.a_max, L.j_max, L.pw, t_sample, v_0, a_0, q, r, dir, mc);
#pragma unroll
    for (int k = 0; k < 7; ++k) out[k] = r[k];
    out[7] = q;
    out[8] = dir;
    out[11] = (double)mc.flags;
}

template <int SEM>
__global__ void __launch_bounds__(64) k_single_opt_switch(int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0, double a_0,
                                    double v_drive, double* io)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();
    const JointLimits L = load_limits(lim, joint);
    double t[7] = {io[0], io[1], io[2], io[3], io[4], io[5], io[6]};
    double dir = 0.0;
    int mod = 0;
    MatlabCtx mc;
    const bool ok = opt_switch_times<true, SEM>(L.a_max, L.j_max, L.v_max, L.pw, t_sample, q_goal, q_0, v_0, a_0, v_drive, t, dir, mod, mc) == kOptTrue;
#pragma unroll
    for (int k = 0; k < 7; ++k) io[k] = t[k];
    io[7] = dir;
    io[8] = (double)mod;
    io[9] = ok ? 1.0 : 0.0;
    io[11] = (double)mc.flags;
}

template <int SEM>
__global__ void __launch_bounds__(64) k_single_time_scaling(int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0, double a_0,
                                      double dir, double tr, double* io)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();
    const JointLimits L = load_limits(lim, joint);
    double ts[7] = {io[0], io[1], io[2], io[3], io[4], io[5], io[6]};
    double vd;
    int mod = 0, which = 0;
    MatlabCtx mc;
    const bool acc = time_scaling_full<SEM>(L, t_sample, q_goal, q_0, v_0, a_0, dir, tr, vd, ts, mod, which, mc);
#pragma unroll
    for (int k = 0; k < 7; ++k) io[k] = ts[k];
    io[7] = vd;
    io[8] = (double)mod;
    io[9] = acc ? 1.0 : 0.0;
    io[10] = (double)which;
    io[11] = (double)mc.flags;
}

// MATLAB's roots() (ltp_roots_matlab.hpp) for the tests of the MATLAB-semantics mode
__global__ void __launch_bounds__(64)
k_roots_matlab(long long n, int degree, const double* __restrict__ coef, double* __restrict__ re, double* __restrict__ im,
               int* __restrict__ nroots, int* __restrict__ status)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double c[mr::kMaxN + 1], r[mr::kMaxN], m[mr::kMaxN];
    for (int k = 0; k <= degree; ++k) c[k] = coef[i * (degree + 1) + k];
    int nr = 0;
    status[i] = mr::roots(c, degree, r, m, nr);
    nroots[i] = nr;
    for (int k = 0; k < degree; ++k) { re[i * degree + k] = r[k]; im[i * degree + k] = m[k]; }
}

// device arithmetic probes: tests compare these with the host's libm bit for bit
__global__ void k_math_probe(long long n, const double* x, const double* y, double* out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y[i];
    double* o = out + i * 8;
    o[0] = a / b;
    o[1] = dsqrt(dabs(a));
    o[2] = pw3_exact(a);
    o[3] = pw4_exact(a);
    o[4] = pw6_exact(a);
    o[5] = dfloor(a / b);
    o[6] = dceil(a / b);
    o[7] = a * b + a;
}

// the restated glibc pow (ltp_libm_pow.hpp, pow rule LTP_POW_LIBM) on arbitrary (x, y): tests compare it with the host's libm bit for bit
__global__ void k_libm_pow_probe(long long n, const double* x, const double* y, double* out)
{
    libm::stage_tables();
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = libm::pow(x[i], y[i]);
}

template <int N>
LTP_DEV double probe_root(const double* c)
{
    double p[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) p[i] = c[i];
    return smallest_positive_real_root<N>(p);
}

// one kernel per degree: with the three solvers in one kernel its register allocation is the largest solver's plus the others'
// live ranges, and the probe would not time what a candidate wave of k_scaling_slow runs
template <int N>
__global__ void __launch_bounds__(64)
k_roots_probe(long long n, const double* coef, double* root)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    root[i] = probe_root<N>(coef + i * 7);
}

// roots<T>() of the reference's long_term_planner/roots.h:22-34 for polynomial i: all eigenvalues of the companion matrix,
// in Eigen's output order (companion_eigenvalues). coef: [n][degree + 1] highest coefficient first; re, im: [n][degree].
template <int N, typename R>
LTP_DEV void roots_all_one(const R* c, R* re, R* im)
{
    R p[N + 1], r[N], m[N];
#pragma unroll
    for (int i = 0; i <= N; ++i) p[i] = c[i];
    companion_eigenvalues<N, R>(p, r, m);
#pragma unroll
    for (int i = 0; i < N; ++i) { re[i] = r[i]; im[i] = m[i]; }
}

template <typename R>
__global__ void k_roots_all(long long n, int degree, const R* coef, R* re, R* im)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const R* c = coef + i * (degree + 1);
    R* r = re + i * degree;
    R* m = im + i * degree;
    switch (degree) {
    case 1: roots_all_one<1, R>(c, r, m); break;
    case 2: roots_all_one<2, R>(c, r, m); break;
    case 3: roots_all_one<3, R>(c, r, m); break;
    case 4: roots_all_one<4, R>(c, r, m); break;
    case 5: roots_all_one<5, R>(c, r, m); break;
    case 6: roots_all_one<6, R>(c, r, m); break;
    case 7: roots_all_one<7, R>(c, r, m); break;
    default: roots_all_one<8, R>(c, r, m); break;
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
void launch_roots_all(hipStream_t s, long long n, int degree, bool f32, const void* coef, void* re, void* im)
{
    if (n <= 0) return;
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    if (f32) hipLaunchKernelGGL(k_roots_all<float>, grid, block, 0, s, n, degree, (const float*)coef, (float*)re, (float*)im);
    else hipLaunchKernelGGL(k_roots_all<double>, grid, block, 0, s, n, degree, (const double*)coef, (double*)re, (double*)im);
}

void launch_generate(hipStream_t s, long long n, int dof, Limits lim, unsigned long long seed, long long first_query,
                     double* q_goal, double* q_0, double* v_0, double* a_0, long long sq, long long sj)
{
    if (n <= 0) return;
    const long long total = n * dof;
    hipLaunchKernelGGL(k_generate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n, dof, lim, seed, first_query,
                       q_goal, q_0, v_0, a_0, sq, sj);
}

void launch_limit_powers(hipStream_t s, int dof, const double* a_max, const double* j_max, double* out_exact, double* out_libm)
{
    if (dof <= 0) return;
    hipLaunchKernelGGL(k_limit_powers<0>, dim3(1), dim3(64), 0, s, dof, a_max, j_max, out_exact);
    hipLaunchKernelGGL(k_limit_powers<kPowLibm>, dim3(1), dim3(64), 0, s, dof, a_max, j_max, out_libm);
}

void launch_check_inputs(hipStream_t s, int dof, Limits lim, const double* q_0, const double* v_0, const double* a_0, int* ok, int variant)
{
    dispatch_variant(variant & 1, [&](auto v) {
        hipLaunchKernelGGL(k_check_inputs<decltype(v)::value>, dim3(1), dim3(1), 0, s, dof, lim, q_0, v_0, a_0, ok);
    });
}
void launch_single_opt_braking(hipStream_t s, int joint, double t_sample, Limits lim, double v_0, double a_0, double* out10, int variant)
{
    dispatch_variant(variant, [&](auto v) {
        hipLaunchKernelGGL(k_single_opt_braking<decltype(v)::value>, dim3(1), dim3(1), 0, s, joint, t_sample, lim, v_0, a_0, out10);
    });
}
void launch_roots_matlab(hipStream_t s, long long n, int degree, const double* coef, double* re, double* im, int* nroots, int* status)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_roots_matlab, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, n, degree, coef, re, im, nroots, status);
}
void launch_single_opt_switch(hipStream_t s, int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0,
                              double a_0, double v_drive, double* io10, int variant)
{
    dispatch_variant(variant, [&](auto v) {
        constexpr int SEM = decltype(v)::value;
        hipLaunchKernelGGL(k_single_opt_switch<SEM>, dim3(1), dim3(1), 0, s, joint, t_sample, lim, q_goal, q_0,
                           v_0, a_0, v_drive, io10);
    });
}
void launch_single_time_scaling(hipStream_t s, int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0,
                                double a_0, double dir, double t_required, double* out11, int variant)
{
    dispatch_variant(variant, [&](auto v) {
        constexpr int SEM = decltype(v)::value;
        hipLaunchKernelGGL(k_single_time_scaling<SEM>, dim3(1), dim3(1), 0, s, joint, t_sample, lim, q_goal, q_0,
                           v_0, a_0, dir, t_required, out11);
    });
}
void launch_libm_pow_probe(hipStream_t s, long long n, const double* x, const double* y, double* out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_libm_pow_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, y, out);
}
void launch_math_probe(hipStream_t s, long long n, const double* x, const double* y, double* out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_math_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, y, out);
}
void launch_roots_probe(hipStream_t s, long long n, int degree, const double* coef, double* root)
{
    if (n <= 0) return;
    const dim3 grid((unsigned)((n + 63) / 64)), block(64);
    if (degree == 4) hipLaunchKernelGGL(k_roots_probe<4>, grid, block, 0, s, n, coef, root);
    else if (degree == 5) hipLaunchKernelGGL(k_roots_probe<5>, grid, block, 0, s, n, coef, root);
    else hipLaunchKernelGGL(k_roots_probe<6>, grid, block, 0, s, n, coef, root);
}

}  // namespace ltp
