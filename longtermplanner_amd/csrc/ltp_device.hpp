// ltp_device.hpp — small device-side helpers shared by the kernel translation units.
#pragma once
#include "ltp_kernels.hpp"
#include "ltp_profile.hpp"

namespace ltp {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// 16-byte store unit of an output row: 2 doubles or 4 floats
template <typename T> struct OutVec;
template <> struct OutVec<double> { typedef double2_t type; static constexpr int N = 2; };
template <> struct OutVec<float> { typedef float4_t type; static constexpr int N = 4; };

LTP_DEV LimPow load_limit_powers(const Limits& lim, int j)
{
    const double* w = lim.pw + (long long)j * kLimPowN;
    return LimPow{w[0], w[1], w[2], w[3], w[4], w[5], w[6]};
}

LTP_DEV JointLimits load_limits(const Limits& lim, int j)
{
    JointLimits L;
    L.q_min = lim.q_min[j];
    L.q_max = lim.q_max[j];
    L.v_max = lim.v_max[j];
    L.a_max = lim.a_max[j];
    L.j_max = lim.j_max[j];
    L.pw = load_limit_powers(lim, j);
    return L;
}

// (int)ceil(t[6]/Ts) + 1 of one joint (cc:718), or -1 if any of its switching times is not finite or the length does
// not fit an int (both DEFINED here: the reference converts out-of-range doubles to int, which is undefined)
LTP_DEV int joint_len(const double (&t)[7], double t_sample)
{
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 7; ++k) finite = finite && dfinite(t[k]);
    const double len = dceil(t[6] / t_sample) + 1.0;
    return (finite && len < 2147483647.0) ? (int)len : -1;
}

// Samples stored per row: every rows.stride-th sample (0, stride, 2*stride, ...), at most rows.max_samples of them.
// {0, 1} stores whole trajectories, which is the reference's behaviour.
LTP_HD int stored_len(int len, RowSpec rows)
{
    if (len <= 0) return 0;
    const int st = rows.stride > 1 ? rows.stride : 1;
    const int cnt = (len + st - 1) / st;
    return (rows.max_samples > 0 && cnt > rows.max_samples) ? rows.max_samples : cnt;
}

LTP_DEV unsigned long long plan_size(int len, int dof)
{
    if (len <= 0) return 0ull;
    const unsigned long long stride = ((unsigned long long)len + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    return 4ull * (unsigned long long)dof * stride;
}

}  // namespace ltp
