// libm_pow_test.cc — the restated glibc pow (longtermplanner_amd/csrc/ltp_libm_pow.hpp, pow rule LTP_POW_LIBM) against the
// installed libm's pow, on the host. Plain g++, no GPU, no HIP: the header is host/device-portable and the device build runs the
// same operations (tests/test_gpu_parity.py compares the device's powers with libm's through ltp_debug_arith).
//   usage: libm_pow_test <millions of inputs per class> <seed>      exit code 0 = every result bit-identical
// Build flags matter: -ffp-contract=off (every fusion of the restated build is written out with __builtin_fma; nothing else may
// fuse) and -mfma (so that __builtin_fma is the instruction, not a libm call; the results are the same either way).
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <initializer_list>

#include "../../longtermplanner_amd/csrc/ltp_libm_pow.hpp"

static uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static double u01(uint64_t& s) { return (double)(splitmix(s) >> 11) * 0x1p-53; }
static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static double from_bits(uint64_t u) { double x; memcpy(&x, &u, 8); return x; }
static bool same(double a, double b) { return bits(a) == bits(b) || (a != a && b != b); }

static long long g_bad = 0, g_n = 0, g_differs_from_exact = 0;
// volatile function pointer: the call is libm's pow at run time, never folded by the compiler
static double (*volatile libm_pow)(double, double) = static_cast<double (*)(double, double)>(&pow);

static void check(double x, double y)
{
    const double want = libm_pow(x, y);
    const double got = ltp::libm::pow(x, y);
    ++g_n;
    if (!same(want, got)) {
        if (g_bad < 10) fprintf(stderr, "MISMATCH pow(%a, %a): libm %a, restated %a\n", x, y, want, got);
        ++g_bad;
    }
    // the fixed-exponent form the kernels call (pw3 / pw4 / pw6 / pw_half of ltp_math.hpp)
    double fixed = want;
    if (y == 0.5) fixed = ltp::libm::pow_fixed<1>(x);
    else if (y == 3.0) fixed = ltp::libm::pow_fixed<6>(x);
    else if (y == 4.0) fixed = ltp::libm::pow_fixed<8>(x);
    else if (y == 6.0) fixed = ltp::libm::pow_fixed<12>(x);
    else return;
    ++g_n;
    if (!same(want, fixed)) {
        if (g_bad < 10) fprintf(stderr, "MISMATCH pow_fixed(%a, %a): libm %a, restated %a\n", x, y, want, fixed);
        ++g_bad;
    }
}

int main(int argc, char** argv)
{
    const long long per = (argc > 1 ? atoll(argv[1]) : 2) * 1000000ll;
    uint64_t s = argc > 2 ? strtoull(argv[2], nullptr, 0) : 12345;
    static const double ys[5] = {0.5, 2.0, 3.0, 4.0, 6.0};
    // class 1: the planner's powers — x a time, acceleration, jerk, velocity: log-uniform magnitude 1e-9 .. 1e6, both signs
    for (long long i = 0; i < per; ++i) {
        const double mag = std::exp((u01(s) * 15.0 - 9.0) * 2.302585092994046);
        const double x = (splitmix(s) & 1) ? -mag : mag;
        for (double y : ys) check(x, y);
        // how often libm's pow is NOT the correctly rounded power (the reason this rule exists)
        const long double e3 = (long double)x * x * x;
        if (bits(libm_pow(x, 3.0)) != bits((double)e3)) ++g_differs_from_exact;
    }
    // class 2: any finite x bit pattern (subnormals, huge, negative) with the planner's exponents: under/overflow paths
    for (long long i = 0; i < per; ++i) {
        const double x = from_bits(splitmix(s));
        for (double y : ys) check(x, y);
    }
    // class 3: arbitrary (x, y): y log-uniform 1e-3 .. 1e3, both signs, integers and non-integers; x positive and negative
    for (long long i = 0; i < per; ++i) {
        const double x = std::exp((u01(s) * 40.0 - 20.0) * 2.302585092994046) * ((splitmix(s) & 3) ? 1.0 : -1.0);
        double y = std::exp((u01(s) * 6.0 - 3.0) * 2.302585092994046) * ((splitmix(s) & 1) ? 1.0 : -1.0);
        if (splitmix(s) & 1) y = std::nearbyint(y);
        check(x, y);
        check(x, from_bits(splitmix(s)));                          // any y bit pattern at all (tiny, huge, NaN, inf)
    }
    // class 4: edge values, every pair
    for (double x : {0.0, -0.0, 1.0, -1.0, (double)INFINITY, -(double)INFINITY, (double)NAN, 0x1p-1074, -0x1p-1074, 0x1p-1022, -0x1.8p-1030, 0x1.fffffffffffffp1023,
                     -0x1.fffffffffffffp1023, 0x1.fffffffffffffp-1, 0x1.0000000000001p0, -0x1.0000000000001p0, 0x1p170, 0x1p171, 0x1p256, -0x1p256, 0x1p340, -0x1p342,
                     0x1p-170, 0x1p-180, -0x1p-256, 0x1p-341, -0x1p-358, 0x1p-359, 0x1p-400, 1e300, -1e300, 1e-300, -1e-300})
        for (double y : ys) check(x, y);
    // the fixed exponents where exp leaves its main path: y log2|x| in +-[700, 1100] covers exp's specialcase range
    // (|y ln x| in [2^9, 2^10)), the overflow threshold (1024) and the subnormal results down to underflow (-1075)
    for (long long i = 0; i < per / 4; ++i) {
        const double yy = ys[splitmix(s) % 5];
        const double l2 = (u01(s) * 400.0 + 700.0) * ((splitmix(s) & 1) ? 1.0 : -1.0);
        const double x = std::exp2(l2 / yy) * ((splitmix(s) & 1) ? 1.0 : -1.0);
        if (std::isfinite(x) && x != 0.0) check(x, yy);
    }
    static const double edge[] = {0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, INFINITY, -INFINITY, NAN, 0x1p-1074, -0x1p-1074,
                                  0x1p-1022, 0x1.fffffffffffffp1023, -0x1.fffffffffffffp1023, 0x1p-65, 0x1p63, 0x1p-66, 0x1p64, 1e-300, 1e300,
                                  0x1.fffffffffffffp-1, 0x1.0000000000001p0, 709.78, -745.13, 1074.0, -1075.0, 1023.9999, 0x1p52, 0x1p53 + 2, 0x1p53 - 1};
    for (double x : edge)
        for (double y : edge) check(x, y);
    // class 5: results in the subnormal / overflow range (exp's specialcase): x^y with y log2|x| near +-1022 .. +-1075
    for (long long i = 0; i < per / 4; ++i) {
        const double x = std::exp((u01(s) * 20.0 - 10.0) * 2.302585092994046);
        const double target = (u01(s) * 120.0 + 960.0) * ((splitmix(s) & 1) ? 1.0 : -1.0);
        const double y = target / std::log2(x);
        if (std::isfinite(y)) check(x, y);
    }
    printf("{\"inputs\": %lld, \"mismatches\": %lld, \"libm_pow3_not_correctly_rounded_frac\": %.4f}\n", g_n, g_bad,
           (double)g_differs_from_exact / (double)per);
    return g_bad ? 1 : 0;
}
