// ltp_math.hpp — per-lane binary64 helpers for the gfx950 planner kernels.
//
// The reference computes everything in scalar IEEE double with libm pow()/sqrt()
// (reference src/long_term_planner.cc, e.g. :133, :175-181, :202-223). To stay
// within 1e-9 of it across discontinuous branch tests, device arithmetic must be
// the same operations in the same order:
//   * this translation unit is built with -ffp-contract=off and without fast-math,
//     so a*b+c is two roundings exactly as on the host; '/' and sqrt() lower to the
//     correctly rounded f64 sequences on gfx950;
//   * pow(x, n) for the integer exponents the reference uses (3, 4, 6; gcc folds pow(x, 2) to x * x) and its one
//     pow(x, 1.0 / 2) follow one of two POW RULES (ltp_set_pow_rule):
//       LTP_POW_LIBM (default)   glibc's pow restated operation for operation (ltp_libm_pow.hpp): the bits a reference built
//                                with gcc + glibc (>= 2.28, FMA host) computes;
//       LTP_POW_EXACT            one rounding of the exact product (error-free products through fma), sqrt for 1/2:
//                                what a correctly rounded pow returns; within 1 ulp of any libm; ~1/4 fewer stage-kernel cycles.
//     The rule rides in the template parameter SEM of ltp_profile.hpp next to the semantics: SEM = semantics | kPowLibm.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ltp_run_tables.hpp"   // LTP_DEV / LTP_HD, kSemCpp / kSemMatlab
#include "ltp_libm_pow.hpp"

namespace ltp {


constexpr double kInf = __builtin_huge_val();

// SEM = kSemCpp | kSemMatlab (bit 0: whose translation, include/ltp_run_tables.hpp), optionally | kPowLibm (bit 1: the pow rule).
// Everything below the stage kernels (runs, samplers, consumers) has no powers and only ever sees bit 0.
constexpr int kPowLibm = 2;
constexpr bool sem_matlab(int sem) { return (sem & 1) != 0; }
constexpr bool sem_libm(int sem) { return (sem & kPowLibm) != 0; }

LTP_DEV double dabs(double x) { return __builtin_fabs(x); }
LTP_DEV bool dfinite(double x) { return dabs(x) < kInf; }   // false for inf and NaN
LTP_DEV bool disnan(double x) { return x != x; }

// reference include/long_term_planner/long_term_planner.h:54-56
LTP_DEV int sgn(double v) { return (0.0 < v) - (v < 0.0); }

// pow(x,2): a single product is already the correctly rounded square
LTP_DEV double pw2(double x) { return x * x; }

// a*b = hi + lo exactly (lo via fused multiply-add)
LTP_DEV void two_prod(double a, double b, double& hi, double& lo)
{
    hi = a * b;
    lo = __builtin_fma(a, b, -hi);
}

// pow(x,3) rounded once
LTP_DEV double pw3_exact(double x)
{
    double h, l, p, e;
    two_prod(x, x, h, l);
    two_prod(h, x, p, e);
    double r = p + (e + l * x);
    return dfinite(p) ? r : h * x;
}

// pow(x,4) rounded once
LTP_DEV double pw4_exact(double x)
{
    double h, l, p, e;
    two_prod(x, x, h, l);
    two_prod(h, h, p, e);
    double r = p + (e + 2.0 * (h * l));
    return dfinite(p) ? r : h * h;
}

// pow(x,6) rounded once
LTP_DEV double pw6_exact(double x)
{
    double h, l, p3, e3, p, e;
    two_prod(x, x, h, l);
    two_prod(h, x, p3, e3);
    e3 = e3 + l * x;                 // x^3 = p3 + e3
    two_prod(p3, p3, p, e);
    double r = p + (e + 2.0 * (p3 * e3));
    return dfinite(p) ? r : (h * x) * (h * x);
}

LTP_DEV double dsqrt(double x) { return __builtin_sqrt(x); }

// the reference's pow(x, 3 | 4 | 6) and pow(x, 1.0 / 2) under the pow rule of SEM
template <int SEM> LTP_DEV double pw3(double x) { if constexpr (sem_libm(SEM)) return libm::pow_fixed<6>(x); else return pw3_exact(x); }
template <int SEM> LTP_DEV double pw4(double x) { if constexpr (sem_libm(SEM)) return libm::pow_fixed<8>(x); else return pw4_exact(x); }
template <int SEM> LTP_DEV double pw6(double x) { if constexpr (sem_libm(SEM)) return libm::pow_fixed<12>(x); else return pw6_exact(x); }
template <int SEM> LTP_DEV double pw_half(double x) { if constexpr (sem_libm(SEM)) return libm::pow_fixed<1>(x); else return dsqrt(x); }
LTP_DEV double dfloor(double x) { return __builtin_floor(x); }
LTP_DEV double dceil(double x) { return __builtin_ceil(x); }
LTP_DEV double dmax(double a, double b) { return a < b ? b : a; }   // std::max semantics
LTP_DEV double dmin(double a, double b) { return b < a ? b : a; }   // std::min semantics

}  // namespace ltp
