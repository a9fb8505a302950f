// ltp_host_probe.cc — which pow rule reproduces the libm THIS process runs on (C ABI: ltp_host_libm_pow_rule).
//
// The reference forms pow(x, 3 | 4 | 6) and pow(x, 1.0 / 2) with the C library's pow (src/long_term_planner.cc:125-331, 378-621).
// LTP_POW_LIBM restates ONE libm: glibc >= 2.28 in the build it selects on x86-64 hosts with FMA (ltp_libm_pow.hpp). A reference
// linked against another libm (musl, an older glibc, a host without FMA) differs from that in the last bit of about one power in
// a thousand, and its records then differ from the device's by up to ~5e-11 s in ~2 plans per million. A caller
// cannot see that from the API — so this entry point compares, in the caller's own process, the installed pow() with both rules.
//
// Plain C++ (g++, no HIP): ltp_libm_pow.hpp's host form. Built with -ffp-contract=off like everything in csrc/, because the
// header writes out every fusion of the build it restates — which is why this is a library function and not a header-only probe:
// in a user's translation unit the compiler's default contraction would change the restated arithmetic.
#include <cmath>
#include <cstdint>
#include <cstring>
#include "../../include/ltp_hip.h"
#include "ltp_libm_pow.hpp"

namespace {

// the exact rule (ltp_math.hpp pw3_exact / pw4_exact / pw6_exact, dsqrt): one rounding of the exact power
inline void two_prod(double a, double b, double& hi, double& lo) { hi = a * b; lo = std::fma(a, b, -hi); }
inline double exact_rule(double x, int y2)
{
    double h, l, p, e;
    switch (y2) {
    case 1: return std::sqrt(x);
    case 6: { two_prod(x, x, h, l); two_prod(h, x, p, e); const double r = p + (e + l * x); return std::isfinite(p) ? r : h * x; }
    case 8: { two_prod(x, x, h, l); two_prod(h, h, p, e); const double r = p + (e + 2.0 * (h * l)); return std::isfinite(p) ? r : h * h; }
    default: {
        double p3, e3;
        two_prod(x, x, h, l); two_prod(h, x, p3, e3); e3 = e3 + l * x;
        two_prod(p3, p3, p, e);
        const double r = p + (e + 2.0 * (p3 * e3));
        return std::isfinite(p) ? r : (h * x) * (h * x);
    }
    }
}

inline uint64_t splitmix64(uint64_t& s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

inline bool same_bits(double a, double b) { return std::memcmp(&a, &b, sizeof a) == 0; }

}  // namespace

extern "C" int ltp_host_libm_pow_rule(long long probes, long long* mismatches_libm, long long* mismatches_exact)
{
    if (probes <= 0) probes = 1 << 18;           // one power in ~1 000 tells the rules apart: 2^18 probes see ~250 such powers
    // through a volatile pointer: the compiler must call the installed libm, not fold or expand pow itself
    double (*volatile host_pow)(double, double) = static_cast<double (*)(double, double)>(&std::pow);
    static const int y2s[4] = {6, 8, 12, 1};     // pow(x, 3), (x, 4), (x, 6), (x, 1.0 / 2): the planner's exponents
    uint64_t seed = 0x6c74705f70726f62ull;
    long long nl = 0, ne = 0;
    for (long long i = 0; i < probes; ++i) {
        // planner-sized magnitudes: |x| log-uniform in [2^-20, 2^14), either sign for the integer exponents
        const uint64_t u = splitmix64(seed);
        const double m = 1.0 + (double)(u >> 12) * 0x1p-52;
        const int ex = (int)((u >> 4) & 0xff) % 34 - 20;
        const int y2 = y2s[i & 3];
        double x = std::ldexp(m, ex);
        if (y2 != 1 && (u & 1)) x = -x;
        const double ref = host_pow(x, 0.5 * y2);
        double restated;
        switch (y2) {
        case 6: restated = ltp::libm::pow_fixed<6>(x); break;
        case 8: restated = ltp::libm::pow_fixed<8>(x); break;
        case 12: restated = ltp::libm::pow_fixed<12>(x); break;
        default: restated = ltp::libm::pow_fixed<1>(x); break;
        }
        nl += same_bits(ref, restated) ? 0 : 1;
        ne += same_bits(ref, exact_rule(x, y2)) ? 0 : 1;
    }
    if (mismatches_libm) *mismatches_libm = nl;
    if (mismatches_exact) *mismatches_exact = ne;
    return nl == 0 ? LTP_POW_LIBM : (ne == 0 ? LTP_POW_EXACT : -1);
}
