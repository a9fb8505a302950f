// ltp_libm_pow.hpp — glibc's pow(double, double) restated operation for operation (pow rule LTP_POW_LIBM).
//
// Why: the reference forms its powers with libm's pow — pow(x, 3 | 4 | 6) and one pow(x, 1.0 / 2)
// (/root/reference/src/long_term_planner.cc:125-331, 378-621; gcc folds pow(x, 2) to x * x, never the others). glibc's pow is
// within 0.52 ulp but not correctly rounded, so a reference built with gcc + glibc differs in the last bit of about one power in a thousand
// from the correctly rounded products of ltp_math.hpp (pw3 / pw4 / pw6), and timeScaling's cancelling v_drive formulas
// (cc:378-446) amplify that into up to 4.5e-11 s of a switching time and, through |dt| * j_max / Ts (cc:768-807), into jerk
// samples beyond 1e-9 in ~2 plans per million. With this rule the device computes those powers exactly as the
// reference's libm does, bit for bit.
//
// What is restated: glibc >= 2.28's pow (sysdeps/ieee754/dbl-64/e_pow.c, Szabolcs Nagy's algorithm, the same as ARM
// optimized-routines math/pow.c; not under /root/reference — a system library): log(x) through a 128-entry table of (1/c, log c)
// and a degree-7 polynomial in r = z/c - 1, carried as hi + lo; y * log(x) as ehi + elo; exp through a 128-entry table of
// 2^(k/128) and a degree-5 polynomial. Tables: ltp_libm_pow_tables.inc, GENERATED from the rules the sources state
// (tools/gen_libm_pow_tables.py); polynomial coefficients and ln2 splits: quoted below from the published sources.
// glibc's x86-64 libm picks, on every CPU with FMA and AVX2 (every x86 server since 2013, all hosts of this pool), the build of
// that file compiled with -mfma (sysdeps/x86_64/fpu/multiarch/e_pow-fma.c), in which __FP_FAST_FMA selects the fma forms of
// log_inline / pow AND gcc's default -ffp-contract=fast fuses every a * b + c of one expression: the fusions below are written out
// with __builtin_fma exactly where that build has them (read off Ubuntu GLIBC 2.35-0ubuntu3.11's __pow_fma; a host without FMA
// runs the unfused variant and differs from this in rare last bits, like two hosts of the reference differ from each other).
//
// Pinned: tests/cpp/libm_pow_test.cc compiles this header with g++ and compares it with the installed libm's pow on 10^8 random
// (x, y) per run, y in {0.5, 2, 3, 4, 6} and arbitrary, incl. negative / subnormal / huge / non-finite x and results in the
// subnormal and overflow ranges: bit-identical everywhere (tests/test_libm_pow.py; a soak of 2e10 inputs is in profiles/).
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define LTP_LIBM_FN __device__ __forceinline__
#define LTP_LIBM_RARE __device__ __attribute__((noinline))
#define LTP_LIBM_TAB __device__ static const
#else
#define LTP_LIBM_FN static inline
#define LTP_LIBM_RARE static __attribute__((noinline))
#define LTP_LIBM_TAB static const
#endif

namespace ltp {
namespace libm {

#include "ltp_libm_pow_tables.inc"

typedef unsigned long long u64;

// Where the kernels read the two tables from. On the device every block keeps a copy in LDS (5 KB: a kernel that forms libm powers calls
// stage_tables() once, all threads, before its first power): a lane's table row is a data-dependent address, i.e. 64 different lines
// per wave instruction, which LDS serves at a fraction of the latency of the vector L1. On the host (tests) the arrays themselves.
#if defined(__HIPCC__) && !defined(LTP_LIBM_GLOBAL_TABLES)
LTP_LIBM_FN double* lds_log_tab() { __shared__ double t[128 * 3]; return t; }
LTP_LIBM_FN u64* lds_exp_tab() { __shared__ u64 t[256]; return t; }
__device__ __forceinline__ void stage_tables()
{
    const int threads = (int)(blockDim.x * blockDim.y * blockDim.z);
    const int tid = ((int)threadIdx.z * (int)blockDim.y + (int)threadIdx.y) * (int)blockDim.x + (int)threadIdx.x;
    double* lt = lds_log_tab();
    u64* et = lds_exp_tab();
    for (int i = tid; i < 128 * 3; i += threads) lt[i] = (&kPowLogTab[0][0])[i];
    for (int i = tid; i < 256; i += threads) et[i] = kExpTab[i];
    __syncthreads();
}
#define LTP_LIBM_LOG_ROW(i, k) lds_log_tab()[(i) * 3 + (k)]
#define LTP_LIBM_EXP(i) lds_exp_tab()[(i)]
#else
#if defined(__HIPCC__)
__device__ __forceinline__ void stage_tables() {}      // (-DLTP_LIBM_GLOBAL_TABLES: A/B builds that read the arrays in global memory)
#endif
#define LTP_LIBM_LOG_ROW(i, k) kPowLogTab[(i)][(k)]
#define LTP_LIBM_EXP(i) kExpTab[(i)]
#endif

LTP_LIBM_FN u64 bits(double x) { return __builtin_bit_cast(u64, x); }
LTP_LIBM_FN double from_bits(u64 u) { return __builtin_bit_cast(double, u); }
LTP_LIBM_FN unsigned top12(double x) { return (unsigned)(bits(x) >> 52); }

// e_pow_log_data.c: ln2 split and the polynomial of log1p(r) - r, "scaled to match the scaling during evaluation"
constexpr double kLn2hi = 0x1.62e42fefa3800p-1, kLn2lo = 0x1.ef35793c76730p-45;
constexpr double kA0 = -0x1p-1, kA1 = 0x1.555555555556p-2 * -2, kA2 = -0x1.0000000000006p-2 * -2, kA3 = 0x1.999999959554ep-3 * 4,
                 kA4 = -0x1.555555529a47ap-3 * 4, kA5 = 0x1.2495b9b4845e9p-3 * -8, kA6 = -0x1.0002b8b263fc3p-3 * -8;
// e_exp_data.c (N = 128): N/ln2, the 1.5 * 2^52 shift, -ln2/N split, the polynomial of exp(r) - 1 - r
constexpr double kInvLn2N = 0x1.71547652b82fep0 * 128, kShift = 0x1.8p52, kNegLn2hiN = -0x1.62e42fefa0000p-8, kNegLn2loN = -0x1.cf79abc9e3b3ap-47;
constexpr double kC2 = 0x1.ffffffffffdbdp-2, kC3 = 0x1.555555555543cp-3, kC4 = 0x1.55555cf172b91p-5, kC5 = 0x1.1111167a4d017p-7;
constexpr u64 kOff = 0x3fe6955500000000ull;
constexpr unsigned kSignBias = 0x800u << 7;

// e_pow.c log_inline (the __FP_FAST_FMA form): log(x) = *tail + result for the bit pattern ix of a positive normal x
LTP_LIBM_FN double log_inline(u64 ix, double& tail)
{
    const u64 tmp = ix - kOff;
    const int i = (int)((tmp >> (52 - 7)) % 128);
    const int k = (int)((long long)tmp >> 52);
    const double z = from_bits(ix - (tmp & (0xfffull << 52)));
    const double kd = (double)k;
    const double invc = LTP_LIBM_LOG_ROW(i, 0), logc = LTP_LIBM_LOG_ROW(i, 1), logctail = LTP_LIBM_LOG_ROW(i, 2);
    const double r = __builtin_fma(z, invc, -1.0);                 // exact: 1/c has 9 significant bits
    const double t1 = __builtin_fma(kd, kLn2hi, logc);             // k ln2 + log c + r in two pieces
    const double t2 = t1 + r;
    const double lo1 = __builtin_fma(kd, kLn2lo, logctail);
    const double lo2 = t1 - t2 + r;
    const double ar = kA0 * r;
    const double ar2 = r * ar;
    const double ar3 = r * ar2;
    const double hi = t2 + ar2;
    const double lo3 = __builtin_fma(ar, r, -ar2);
    const double lo4 = t2 - hi + ar2;
    // p = ar3 * (A1 + r A2 + ar2 (A3 + r A4 + ar2 (A5 + r A6))), and lo = lo1 + lo2 + lo3 + lo4 + p with the last two fused
    const double a12 = __builtin_fma(r, kA2, kA1), a34 = __builtin_fma(r, kA4, kA3), a56 = __builtin_fma(r, kA6, kA5);
    const double poly = __builtin_fma(ar2, __builtin_fma(ar2, a56, a34), a12);
    const double lo = __builtin_fma(ar3, poly, lo1 + lo2 + lo3 + lo4);
    const double y = hi + lo;
    tail = hi - y + lo;
    return y;
}

// e_pow.c specialcase: 2^9 <= |x| < 2^10, the scale's exponent is out of range
LTP_LIBM_FN double exp_specialcase(double tmp, u64 sbits, u64 ki)
{
    if ((ki & 0x80000000ull) == 0) {
        sbits -= 1009ull << 52;                                     // k > 0: the exponent of scale may have overflowed by <= 460
        const double scale = from_bits(sbits);
        return 0x1p1009 * __builtin_fma(scale, tmp, scale);
    }
    sbits += 1022ull << 52;                                         // k < 0: care in the subnormal range
    const double scale = from_bits(sbits);
    const double st = scale * tmp;
    double y = scale + st;
    if (__builtin_fabs(y) < 1.0) {
        const double one = y < 0.0 ? -1.0 : 1.0;
        double lo = scale - y + st;
        const double hi = one + y;
        lo = one - hi + y + lo;
        y = (hi + lo) - one;
        if (y == 0.0) y = from_bits(sbits & 0x8000000000000000ull);
    }
    return 0x1p-1022 * y;
}

// e_pow.c exp_inline: exp(x + xtail) with the sign of sign_bias; |xtail| < 2^-8 / N, |xtail| <= |x|
LTP_LIBM_FN double exp_inline(double x, double xtail, unsigned sign_bias)
{
    unsigned abstop = top12(x) & 0x7ff;
    if (abstop - 0x3c9u >= 0x03fu) {                                // top12(0x1p-54) = 0x3c9, top12(0x1p9) = 0x408
        if (abstop - 0x3c9u >= 0x80000000u) {
            const double one = 1.0 + x;                             // tiny: WANT_ROUNDING
            return sign_bias ? -one : one;
        }
        if (abstop >= 0x409u) {                                     // |x| >= 2^10: __math_uflow / __math_oflow
            const double big = (bits(x) >> 63) ? 0x1p-767 : 0x1p769;
            return (sign_bias ? -big : big) * big;
        }
        abstop = 0;                                                 // large x is special-cased below
    }
    const double kd0 = __builtin_fma(kInvLn2N, x, kShift);          // z + Shift, fused in that build
    const u64 ki = bits(kd0);
    const double kd = kd0 - kShift;
    double r = __builtin_fma(kd, kNegLn2loN, __builtin_fma(kd, kNegLn2hiN, x));
    r += xtail;
    const unsigned idx = 2 * (unsigned)(ki % 128);
    const u64 top = (ki + sign_bias) << (52 - 7);
    const double tail = from_bits(kExpTab[idx]);                    // (this general form reads the arrays themselves: it is also what the
    const u64 sbits = kExpTab[idx + 1] + top;                       // out-of-line exp_large runs, where a block's LDS copy is out of reach)
    const double r2 = r * r;
    // tmp = tail + r + r2 (C2 + r C3) + r2 r2 (C4 + r C5), fused from the inside out
    const double tmp = __builtin_fma(__builtin_fma(r, kC5, kC4), r2 * r2, __builtin_fma(__builtin_fma(r, kC3, kC2), r2, tail + r));
    if (abstop == 0) return exp_specialcase(tmp, sbits, ki);
    const double scale = from_bits(sbits);
    return __builtin_fma(scale, tmp, scale);
}

// 0: y is not an integer, 1: odd, 2: even
LTP_LIBM_FN int checkint(u64 iy)
{
    const int e = (int)(iy >> 52 & 0x7ff);
    if (e < 0x3ff) return 0;
    if (e > 0x3ff + 52) return 2;
    if (iy & ((1ull << (0x3ff + 52 - e)) - 1)) return 0;
    if (iy & (1ull << (0x3ff + 52 - e))) return 1;
    return 2;
}

LTP_LIBM_FN bool zeroinfnan(u64 i) { return 2 * i - 1 >= 2 * 0x7ff0000000000000ull - 1; }

// e_pow.c pow. Exceptions and errno are not modelled (the planner reads neither); signalling NaNs are treated as quiet.
LTP_LIBM_FN double pow(double x, double y)
{
    unsigned sign_bias = 0;
    u64 ix = bits(x);
    const u64 iy = bits(y);
    unsigned topx = top12(x);
    const unsigned topy = top12(y);
    if (topx - 0x001u >= 0x7ffu - 0x001u || (topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
        // x < 0x1p-1022 or inf or nan (or negative), or |y| < 0x1p-65 or |y| >= 0x1p63 or nan
        if (zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0;
            if (ix == 0x3ff0000000000000ull) return 1.0;
            if (2 * ix > 2 * 0x7ff0000000000000ull || 2 * iy > 2 * 0x7ff0000000000000ull) return x + y;
            if (2 * ix == 2 * 0x3ff0000000000000ull) return 1.0;
            if ((2 * ix < 2 * 0x3ff0000000000000ull) == !(iy >> 63)) return 0.0;    // |x| < 1 && y == inf, or |x| > 1 && y == -inf
            return y * y;
        }
        if (zeroinfnan(ix)) {
            double x2 = x * x;
            if ((ix >> 63) && checkint(iy) == 1) x2 = -x2;
            return (iy >> 63) ? 1 / x2 : x2;
        }
        if (ix >> 63) {                                             // finite x < 0
            const int yint = checkint(iy);
            if (yint == 0) return (x - x) / (x - x);                // __math_invalid
            if (yint == 1) sign_bias = kSignBias;
            ix &= 0x7fffffffffffffffull;
            topx &= 0x7ff;
        }
        if ((topy & 0x7ff) - 0x3beu >= 0x43eu - 0x3beu) {
            if (ix == 0x3ff0000000000000ull) return 1.0;
            if ((topy & 0x7ff) < 0x3beu) return ix > 0x3ff0000000000000ull ? 1.0 + y : 1.0 - y;   // |y| < 2^-65
            return ((ix > 0x3ff0000000000000ull) == (topy < 0x800)) ? 0x1p769 * 0x1p769 : 0x1p-767 * 0x1p-767;
        }
        if (topx == 0) {                                            // subnormal x: normalise
            ix = bits(from_bits(ix) * 0x1p52);
            ix &= 0x7fffffffffffffffull;
            ix -= 52ull << 52;
        }
    }
    double lo;
    const double hi = log_inline(ix, lo);
    const double ehi = y * hi;
    const double elo = __builtin_fma(y, lo, __builtin_fma(y, hi, -ehi));
    return exp_inline(ehi, elo, sign_bias);
}

// exp_inline where it leaves its main path for good (|x| >= 2^9: the scale's exponent out of range, over- and underflow): one
// out-of-line copy for all call sites of pow_fixed — powers of that size are not planner values, but NaN / inf / huge inputs must
// flow as they do through the reference
LTP_LIBM_RARE double exp_large(double x, double xtail, unsigned sign_bias) { return exp_inline(x, xtail, sign_bias); }

// pow(x, Y2 / 2) for the exponents the planner has — Y2 = 1 (pow(x, 1.0 / 2)), 6, 8, 12 (pow(x, 3 | 4 | 6)) — as pow(x, y) above
// computes it, with the case analysis that a fixed y leaves: the SAME operations on the main path (log_inline, ehi / elo, exp_inline's
// arithmetic), the branches of e_pow.c turned into selects (negative x with an integer y: sign_bias and |x|; subnormal x normalised;
// x = 0 / inf / NaN: x * x with the sign of an odd power; negative x with y = 1/2: NaN; exp's tiny argument: 1 + x) and one rare call.
// tests/cpp/libm_pow_test.cc compares this form, too, with the installed libm's pow on every class of input.
template <int Y2>
LTP_LIBM_FN double pow_fixed(double x)
{
    static_assert(Y2 == 1 || Y2 == 6 || Y2 == 8 || Y2 == 12, "the planner's exponents");
    constexpr double y = 0.5 * Y2;
    constexpr bool kInt = (Y2 % 2) == 0, kOdd = kInt && (Y2 / 2) % 2 == 1;
    const u64 ix0 = bits(x);
    const bool neg = (ix0 >> 63) != 0;
    const u64 iax = ix0 & 0x7fffffffffffffffull;
    const unsigned sign_bias = (kOdd && neg) ? kSignBias : 0u;
    const unsigned top = (unsigned)(iax >> 52);
    // e_pow.c: "Normalize subnormal x so exponent becomes negative" (x = 0 takes the same select and is patched below)
    const u64 isub = (bits(from_bits(iax) * 0x1p52) & 0x7fffffffffffffffull) - (52ull << 52);
    const u64 ix = top == 0 ? isub : iax;
    double lo;
    const double hi = log_inline(ix, lo);
    const double ehi = y * hi;
    const double elo = __builtin_fma(y, lo, __builtin_fma(y, hi, -ehi));
    // exp_inline's main path
    const double kd0 = __builtin_fma(kInvLn2N, ehi, kShift);
    const u64 ki = bits(kd0);
    const double kd = kd0 - kShift;
    double r = __builtin_fma(kd, kNegLn2loN, __builtin_fma(kd, kNegLn2hiN, ehi));
    r += elo;
    const unsigned idx = 2 * (unsigned)(ki % 128);
    const u64 sbits = LTP_LIBM_EXP(idx + 1) + ((ki + sign_bias) << (52 - 7));
    const double tail = from_bits(LTP_LIBM_EXP(idx));
    const double r2 = r * r;
    const double tmp = __builtin_fma(__builtin_fma(r, kC5, kC4), r2 * r2, __builtin_fma(__builtin_fma(r, kC3, kC2), r2, tail + r));
    const double scale = from_bits(sbits);
    double res = __builtin_fma(scale, tmp, scale);
    const unsigned abstop = top12(ehi) & 0x7ff;
    if (abstop - 0x3c9u >= 0x03fu) {
        const double one = 1.0 + ehi;                               // |ehi| < 2^-54 (x next to 1)
        res = sign_bias ? -one : one;
        if (abstop >= 0x408u) res = exp_large(ehi, elo, sign_bias);
    }
    if (top - 1u >= 0x7feu) {                                       // x = 0, subnormal, inf, NaN
        const double x2 = x * x;
        if (zeroinfnan(iax)) res = (kOdd && neg) ? -x2 : x2;
    }
    if (!kInt && neg && iax - 1 < 0x7ff0000000000000ull - 1) res = (x - x) / (x - x);   // finite x < 0, y = 1/2: __math_invalid
    return res;
}

}  // namespace libm
}  // namespace ltp
