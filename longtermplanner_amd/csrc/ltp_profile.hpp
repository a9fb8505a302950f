// ltp_profile.hpp — per-lane 7-phase jerk-limited profile math (binary64).
//
// Device counterparts of the reference's per-joint member functions
// (paths relative to /root/reference):
//   opt_braking        <- LongTermPlanner::optBraking      src/long_term_planner.cc:650-701
//   opt_switch_times   <- LongTermPlanner::optSwitchTimes  src/long_term_planner.cc:82-353
//   v_drive_candidate  <- the eight v_drive formulas of
//                         LongTermPlanner::timeScaling     src/long_term_planner.cc:378-629
//   try_v_drive        <- the acceptance test repeated after each candidate (e.g. cc:398-405)
//   check_inputs_joint <- LongTermPlanner::checkInputs     src/long_term_planner.cc:68-77
//
// One lane = one (query, joint). The joint's limits are wave-uniform in the batch
// kernels (a wave covers 64 queries of one joint), so they sit in SGPRs. Every sum
// keeps the reference's operand order; powers of the inputs are formed once per
// lane (same value as each pow() call in the reference) and reused.
#pragma once
#include "ltp_roots.hpp"
#include "ltp_roots_matlab.hpp"

namespace ltp {

constexpr double kEps = 4e-3;   // cc:96
constexpr double kTol = 0.1;    // cc:370

// Semantics (template parameter SEM of everything below): kSemCpp follows src/long_term_planner.cc — the parity reference and
// the default; kSemMatlab follows the MATLAB original LTPlanner.m wherever the C++ translation diverges from it (SURVEY.md
// §8(f).4, App. C). Every such branch cites the .m lines. What MATLAB does that real arithmetic cannot is defined as in the
// test suite's CPU twin of this mode: a complex intermediate value (sqrt of a negative number, a filtered root with an
// imaginary part below eps) continues with its real part and sets kMatlabComplex, with the |imag(t_rel)| > eps safety test
// (LTPlanner.m:294-297) applied to the imaginary part where it arose; where LTPlanner.m would raise an error (checkInputs,
// an index past the filtered roots, a vector assigned to t_rel(1), NaN / Inf polynomial coefficients) the lane sets
// kMatlabError and the query is rejected.
// (kSemCpp = 0, kSemMatlab = 1: include/ltp_run_tables.hpp)
constexpr int kMatlabComplex = 1, kMatlabError = 2;
struct MatlabCtx {
    int flags = 0;        // kMatlabComplex | kMatlabError, accumulated over a lane's calls
    double imag = 0.0;    // largest imaginary part that entered t_rel in the current optSwitchTimes
};

// sqrt as the formulas of LTPlanner.m see it: MATLAB returns i*sqrt(-x) for x < 0 (the C++: NaN)
template <int SEM>
LTP_DEV double sem_sqrt(double x, MatlabCtx& mc)
{
    if constexpr (sem_matlab(SEM)) {
        if (x < 0.0) {
            const double im = dsqrt(-x);
            mc.flags |= kMatlabComplex;
            mc.imag = dmax(mc.imag, im);
            return 0.0;
        }
    }
    return dsqrt(x);
}

// Powers of one joint's limits, formed ONCE per ltp_set_limits / pow rule by the same device functions the kernels call
// (k_limit_powers, ltp_aux_kernels.hip): tj = a_max / j_max is what optSwitchTimes assigns to t_rel[2], [4], [6] whenever a phase of
// constant acceleration exists (cc:124, 171, 186), i.e. in most lanes; its cube and fourth power, and the cubes / fourth powers of
// a_max and j_max that timeScaling's candidates use, are the same for every query. Under LTP_POW_LIBM a power is ~100 vector
// instructions (glibc's pow restated), and half of the stage kernels' instructions were powers.
struct LimPow {
    double tj, tj3, tj4, am3, am4, jm3, jm4;
};

struct JointLimits {
    double q_min, q_max, v_max, a_max, j_max;
    LimPow pw;
};

// pw3 / pw4 of a value that is USUALLY the joint's tj: when every active lane of the wave holds exactly tj's bits, the stored power
// (the same function of the same bits) is the result and the wave skips the evaluation; otherwise all lanes evaluate.
// (MEMO false: the kernels of the compaction queues, whose lanes hold different joints in per-lane registers — there the stored powers
// would cost six registers per lane in kernels at their register cap and the wave-wide match is rare.)
template <int SEM, bool MEMO = true> LTP_DEV double pw3_tj(double x, const LimPow& P)
{
    if constexpr (!sem_libm(SEM) || !MEMO) return pw3<SEM>(x);        // the exact rule's cube is six operations: nothing to skip
    else if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(unsigned long long, x) != __builtin_bit_cast(unsigned long long, P.tj)) == 0ull) return P.tj3;
    return pw3<SEM>(x);
}
template <int SEM, bool MEMO = true> LTP_DEV double pw4_tj(double x, const LimPow& P)
{
    if constexpr (!sem_libm(SEM) || !MEMO) return pw4<SEM>(x);
    else if (__builtin_amdgcn_ballot_w64(__builtin_bit_cast(unsigned long long, x) != __builtin_bit_cast(unsigned long long, P.tj)) == 0ull) return P.tj4;
    return pw4<SEM>(x);
}

// cc:68-77 for one joint (LTPlanner.m:92-103 has no position limits)
template <int SEM = kSemCpp>
LTP_DEV bool check_inputs_joint(const JointLimits& L, double q_0, double v_0, double a_0)
{
    if constexpr (!sem_matlab(SEM)) {
        if (q_0 < L.q_min || q_0 > L.q_max) return false;
    }
    if (dabs(v_0) > L.v_max || dabs(a_0) > L.a_max) return false;
    if (dabs(v_0 + 0.5 * a_0 * dabs(a_0) / L.j_max) > L.v_max) return false;
    return true;
}

// cc:650-701 (LTPlanner.m:435-484). Writes r[0..2] only.
template <int SEM = kSemCpp, bool MEMO = true>
LTP_DEV void opt_braking(double am, double jm, const LimPow& P, double t_sample, double v_0, double a_0,
                         double& q, double (&r)[7], double& dir, MatlabCtx& mc)
{
    if (v_0 * a_0 > 0.0) {
        dir = -sgn(v_0);
    } else {
        if (dabs(v_0) > 1.0 / 2.0 * pw2(a_0) / jm) dir = -sgn(v_0);
        else dir = -sgn(a_0);
    }
    if (dir < 0.0) {
        a_0 = -a_0;
        v_0 = -v_0;
    }
    r[0] = (am - a_0) / jm;
    r[2] = am / jm;
    r[1] = (-v_0 - 1.0 / 2.0 * r[0] * a_0) / am - 1.0 / 2.0 * (r[0] + r[2]);
    if (r[1] < -t_sample) {
        r[0] = -a_0 / jm + sem_sqrt<SEM>(pw2(a_0) / (2 * pw2(jm)) - v_0 / jm, mc);   // LTPlanner.m:476: complex for a negative argument
        r[2] = r[0] + a_0 / jm;
        r[1] = 0.0;
    }
    q = v_0 * (r[0] + r[1] + r[2]) +
        a_0 * (1.0 / 2.0 * pw2(r[0]) + r[0] * (r[1] + r[2]) + 1.0 / 2.0 * pw2(r[2])) +
        jm * (1.0 / 6.0 * pw3<SEM>(r[0]) + 1.0 / 2.0 * pw2(r[0]) * (r[1] + r[2]) -
              1.0 / 6.0 * pw3_tj<SEM, MEMO>(r[2], P) + 1.0 / 2.0 * r[0] * pw2(r[2])) +
        am * (1.0 / 2.0 * pw2(r[1]) + r[1] * r[2]);
    q = dir * q;
}

// LTPlanner.m:247-250 / 272-275: root = root(abs(imag(root)) < eps); root = root(root >= 0) — MATLAB compares real parts —
// then root(1) (site A), or the whole filtered vector assigned to the scalar t_rel(1) (site B: an error unless exactly one
// root is left). The real part is used; an imaginary part below eps is recorded.
template <int N>
__device__ inline double matlab_filtered_root(const double (&c)[N + 1], bool must_be_single, MatlabCtx& mc)
{
    double re[mr::kMaxN], im[mr::kMaxN];
    int nr = 0;
    const double nan = __builtin_nan("");
    if (mr::roots_n<N>(c, re, im, nr) != 0) { mc.flags |= kMatlabError; return nan; }
    double pick = nan, pick_im = 0.0;
    int kept = 0;
    for (int i = 0; i < nr; ++i) {
        if (dabs(im[i]) < kEps && re[i] >= 0.0) {
            if (kept == 0) { pick = re[i]; pick_im = dabs(im[i]); }
            ++kept;
        }
    }
    if (kept == 0 || (must_be_single && kept != 1)) { mc.flags |= kMatlabError; return nan; }
    if (pick_im != 0.0) {
        mc.flags |= kMatlabComplex;
        mc.imag = dmax(mc.imag, pick_im);
    }
    return pick;
}

// The square of the root a timeScaling candidate uses. C++ (cc:467-627): the smallest positive exactly-real root. LTPlanner.m:
// root(k) BY POSITION in the output of roots() (:346-416); v_drive is then tested with ~imag(v_drive), i.e. the square of a
// complex root must be exactly real, else the candidate is skipped (NaN here has the same effect).
template <int N, int SEM>
__device__ inline double root_squared(const double (&c)[N + 1], int k, MatlabCtx& mc)
{
    if constexpr (sem_matlab(SEM)) {
        double re[mr::kMaxN], im[mr::kMaxN];
        int nr = 0;
        const double nan = __builtin_nan("");
        if (mr::roots_n<N>(c, re, im, nr) != 0 || k > nr) { mc.flags |= kMatlabError; return nan; }
        const double rr = re[k - 1] * re[k - 1] - im[k - 1] * im[k - 1];
        const double ri = 2.0 * (re[k - 1] * im[k - 1]);
        return ri != 0.0 ? nan : rr;
    } else {
        return pw2(smallest_positive_real_root<N>(c));
    }
}

LTP_DEV void cumsum7(const double (&r)[7], double (&t)[7])
{
    double s = r[0];
    t[0] = s;
#pragma unroll
    for (int i = 1; i < 7; ++i) { s = s + r[i]; t[i] = s; }
}
LTP_DEV void zero7(double (&t)[7])
{
#pragma unroll
    for (int i = 0; i < 7; ++i) t[i] = 0.0;
}

// cc:82-353. Returns the reference's bool (kOptFalse / kOptTrue); t is written exactly where the reference
// writes it. With FULL == false the function contains no polynomial solver: a lane that reaches the quartic
// sites (cc:245-337, a few percent of lanes at most) returns kOptDefer having written nothing, and the caller
// re-runs it in a kernel built with FULL == true. That keeps the root finder's registers and code out of the
// kernels every lane runs.
constexpr int kOptFalse = 0, kOptTrue = 1, kOptDefer = 2;

template <bool FULL, int SEM = kSemCpp, bool MEMO = !FULL>
__device__ inline int opt_switch_times(double am, double jm, double vm, const LimPow& P, double t_sample,
                                       double q_goal, double q_0, double v_0, double a_0, double v_drive,
                                       double (&t)[7], double& dir, int& mod, MatlabCtx& mc)
{
    double r[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    mod = 0;
    if constexpr (sem_matlab(SEM)) {
        // LTPlanner.m:131 -> :92-103: checkInputs inside optSwitchTimes, error() on violation (vm = the joint's v_max)
        mc.imag = 0.0;
        if (dabs(v_0) > vm || dabs(a_0) > am || dabs(v_0 + 1.0 / 2.0 * a_0 * dabs(a_0) / jm) > vm) {
            mc.flags |= kMatlabError;
            zero7(t);
            return kOptFalse;
        }
    }
    double q_stop = 0.0;
    opt_braking<SEM, MEMO>(am, jm, P, t_sample, v_0, a_0, q_stop, r, dir, mc);
    const double q_diff = q_goal - (q_0 + q_stop);
    if (dabs(q_diff) < kEps) {
        cumsum7(r, t);
        return kOptTrue;
    }
    dir = sgn(q_diff);
    if (dir < 0.0) {
        v_0 = -v_0;
        a_0 = -a_0;
    }
    const double jm2 = pw2(jm);
    double q_brake = 0.0;
    if (v_0 + 0.5 * a_0 * dabs(a_0) / jm > v_drive) {
        mod = 1;
        double emp;
        opt_braking<SEM, MEMO>(am, jm, P, t_sample, v_0 - v_drive, a_0, q_brake, r, emp, mc);
    } else {
        r[0] = (am - a_0) / jm;
        r[2] = am / jm;
        r[1] = (v_drive - v_0 - 0.5 * r[0] * a_0) / am - 0.5 * (r[0] + r[2]);
        if (r[1] < -kEps) {
            const double root = jm * (v_drive - v_0) + 0.5 * pw2(a_0);
            if (root > 0.0) {
                r[2] = dsqrt(root) / jm;
                r[0] = r[2] - a_0 / jm;
                r[1] = 0.0;
            } else {
                zero7(t);
                return kOptTrue;
            }
        }
    }
    r[4] = am / jm;
    r[6] = r[4];
    r[5] = v_drive / am - 1.0 / 2.0 * (r[4] + r[6]);
    if (r[5] < -kEps) {
        const double root = v_drive / jm;
        if (root > 0.0) {
            r[4] = dsqrt(root);
            r[6] = r[4];
            r[5] = 0.0;
        } else {
            zero7(t);
            return kOptTrue;
        }
    }
    // the cubes of t_rel[0], [2], [4] (= [6], cc:186 / 193) are formed once: the re-solve without a cruise phase below (cc:202-243) uses
    // the same four values again, and a power is the expensive operation here (LimPow)
    double q_part1;
    double r0_3 = 0.0, r2_3 = 0.0;
    if (mod == 1) {
        q_part1 = q_brake + v_drive * (r[0] + r[1] + r[2]);
    } else {
        r0_3 = pw3<SEM>(r[0]);
        r2_3 = pw3_tj<SEM, MEMO>(r[2], P);
        q_part1 = v_0 * (r[0] + r[1] + r[2]) +
                  a_0 * (1.0 / 2.0 * pw2(r[0]) +
                         r[0] * (r[1] + r[2]) +
                         1.0 / 2.0 * pw2(r[2])) +
                  jm * (1.0 / 6.0 * r0_3 +
                        1.0 / 2.0 * pw2(r[0]) * (r[1] + r[2]) -
                        1.0 / 6.0 * r2_3 +
                        1.0 / 2.0 * r[0] * pw2(r[2])) +
                  am * (1.0 / 2.0 * pw2(r[1]) + r[1] * r[2]);
    }
    const double r4_3 = pw3_tj<SEM, MEMO>(r[4], P), r6_3 = r4_3;       // r[6] is r[4] (cc:186, 193)
    const double q_part2 = jm * (1.0 / 6.0 * r6_3 +
                                 1.0 / 2.0 * pw2(r[6]) * (r[5] + r[4]) -
                                 1.0 / 6.0 * r4_3 +
                                 1.0 / 2.0 * r[6] * pw2(r[4])) +
                           am * (1.0 / 2.0 * pw2(r[5]) +
                                 r[5] * r[4]);
    r[3] = ((q_goal - q_0) * dir - q_part1 - q_part2) / v_drive;

    if (r[3] < -kEps) {
        if (mod == 1) {
            zero7(t);
            return sem_matlab(SEM) ? kOptTrue : kOptFalse;   // LTPlanner.m:222-227 returns the zeros like any other result; cc:195-200: false
        }
        const double a2 = pw2(a_0), am2 = pw2(am);
        const double r0_2 = pw2(r[0]), r2_2 = pw2(r[2]);      // mod == 0 here: r0_3, r2_3 are the cubes formed above
        const double r4_2 = pw2(r[4]);
        const double r4_4 = pw4_tj<SEM, MEMO>(r[4], P), r6_4 = r4_4;
        double root = (jm2 * pw4<SEM>(r[0])) / 2 -
                      (jm2 * pw4_tj<SEM, MEMO>(r[2], P)) / 4 +
                      (jm2 * r2_2 * r4_2) / 2 -
                      (jm2 * r4_4) / 4 +
                      (jm2 * r6_4) / 2 +
                      2.0 * jm * a_0 * r0_3 -
                      (2.0 * jm * am * r0_3) / 3 -
                      2.0 * jm * am * r[0] * r2_2 +
                      (2.0 * jm * am * r2_3) / 3 +
                      (2.0 * jm * am * r4_3) / 3 -
                      2.0 * jm * am * r4_2 * r[6] -
                      (2.0 * jm * am * r6_3) / 3 +
                      2.0 * jm * v_0 * r0_2 +
                      2.0 * a2 * r0_2 -
                      2.0 * a_0 * am * r0_2 -
                      2.0 * a_0 * am * r2_2 +
                      4 * a_0 * v_0 * r[0] +
                      2.0 * am2 * r2_2 +
                      2.0 * am2 * r4_2 -
                      4 * am * v_0 * r[0] +
                      4 * dir * (q_goal - q_0) * am +
                      2.0 * pw2(v_0);
        if (root > 0.0) {
            r[5] = -(4 * am * r[4] -
                     2.0 * pw_half<SEM>(root) +
                     jm * r2_2 -
                     jm * r4_2 +
                     2.0 * jm * pw2(r[6])) / (4 * am);
            r[1] = (-v_0 - a_0 * r[0] -
                    1.0 / 2.0 * jm * r0_2 +
                    1.0 / 2.0 * jm * r2_2 +
                    1.0 / 2.0 * jm * pw2(r[6]) -
                    1.0 / 2.0 * jm * r4_2) / am
                   - r[2] + r[5] + r[4];
            r[3] = 0.0;
        } else {
            zero7(t);
            return kOptTrue;
        }

        if (r[5] < -kEps || r[1] < -kEps) {
            if constexpr (!FULL) return kOptDefer;
            // quartic site A (cc:246-261)
            {
                const double c[5] = {
                    12.0,
                    0.0,
                    -24 * a2 + 48 * jm * v_0,
                    48 * dir * jm2 * q_0 -
                        48 * dir * jm2 * q_goal +
                        16 * pw3<SEM>(a_0) - 48 * a_0 * jm * v_0,
                    -3 * pw4<SEM>(a_0) + 12.0 * a2 * jm * v_0 - 12.0 * jm2 * pw2(v_0)};
                if constexpr (sem_matlab(SEM)) {
                    root = matlab_filtered_root<4>(c, false, mc);   // LTPlanner.m:247-250: the first root that passes the filter
                    if (mc.flags & kMatlabError) { zero7(t); return kOptFalse; }
                } else {
                    root = smallest_positive_real_root<4>(c);
                }
            }
            r[0] = (2.0 * pw2(root) - 4 * a_0 * root + a2 - 2.0 * v_0 * jm) / (4 * jm * root);
            r[6] = sem_sqrt<SEM>(4 * jm2 * pw2(r[0]) +
                         8 * a_0 * jm * r[0] +
                         2.0 * a2 +
                         4 * jm * v_0, mc) / (2.0 * jm);
            r[4] = a_0 / jm + r[0] + r[6];
            r[1] = 0.0;
            r[5] = 0.0;

            if (a_0 + r[0] * jm > am) {
                r[0] = (am - a_0) / jm;
                const double n0_2 = pw2(r[0]);
                r[6] = 1.0 / jm * (am / 2 + sem_sqrt<SEM>(
                           9 * am2 + 6 * sem_sqrt<SEM>(
                               -12.0 * am * P.jm3 * pw3<SEM>(r[0]) +
                               9 * a2 * jm2 * n0_2 -
                               18 * a_0 * am * jm2 * n0_2 +
                               9 * am2 * jm2 * n0_2 +
                               36 * a_0 * jm2 * r[0] * v_0 -
                               72.0 * am * dir * jm2 * q_0 +
                               72.0 * am * dir * jm2 * q_goal -
                               36 * am * jm2 * r[0] * v_0 +
                               3 * P.am4 +
                               36 * jm2 * pw2(v_0), mc), mc) / 6.0 - am);
                r[4] = r[6] + am / jm;
                r[1] = -(-jm * pw2(r[4]) -
                         2.0 * jm * r[4] * r[6] +
                         jm * pw2(r[6]) + a_0 * r[0] +
                         am * r[0] +
                         2.0 * am * r[4] +
                         2.0 * am * r[6] +
                         2.0 * v_0) / (2.0 * am);
                r[5] = 0.0;
            }

            if (r[6] * jm > am) {
                // quartic site B (cc:299-321)
                r[6] = am / jm;
                {
                    const double c[5] = {
                        12.0,
                        -24 * am,
                        -12.0 * a2 + 12.0 * am2 + 24 * jm * v_0,
                        0.0,
                        24 * dir * jm2 * q_0 * am -
                            24 * dir * jm2 * q_goal * am +
                            3 * pw4<SEM>(a_0) + 8 * pw3<SEM>(a_0) * am +
                            6 * a2 * am2 -
                            12.0 * a2 * jm * v_0 -
                            24 * a_0 * jm * v_0 * am -
                            12.0 * am2 * jm * v_0 +
                            12.0 * jm2 * pw2(v_0)};
                    if constexpr (sem_matlab(SEM)) {
                        root = matlab_filtered_root<4>(c, true, mc);   // LTPlanner.m:272-275: the filter must leave exactly one root
                        if (mc.flags & kMatlabError) { zero7(t); return kOptFalse; }
                    } else {
                        root = smallest_positive_real_root<4>(c);
                    }
                }
                r[0] = (root - a_0 - am) / jm;
                r[4] = (a_0 + am) / jm + r[0];
                r[5] = (jm2 * pw2(r[0]) +
                        2.0 * jm2 * r[0] * r[4] -
                        jm2 * pw2(r[4]) +
                        2.0 * a_0 * jm * r[0] +
                        2.0 * a_0 * jm * r[4] -
                        am2 +
                        2.0 * jm * v_0) / (2.0 * jm * am);
                r[1] = 0.0;
            }
            r[2] = 0.0;
            r[3] = 0.0;
        }
    }
    if constexpr (sem_matlab(SEM)) {
        // LTPlanner.m:288-303: any(t_rel < -eps) or any(|imag(t_rel)| > eps) zeroes t_rel (no failure); then
        // t_rel = max(0, real(t_rel)), which also turns NaN into 0 (MATLAB's max ignores NaN)
        bool zero_all = mc.imag > kEps;
#pragma unroll
        for (int i = 0; i < 7; ++i) zero_all = zero_all || r[i] < -kEps;
#pragma unroll
        for (int i = 0; i < 7; ++i) r[i] = (zero_all || !(r[i] > 0.0)) ? 0.0 : r[i];
        cumsum7(r, t);
        return kOptTrue;
    }
    // cc:340-348 (the reference's std::cerr diagnostic has no device counterpart)
    bool bad = false;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        if (r[i] < -kEps) bad = true;
        else if (r[i] < 0.0 && r[i] >= -kEps) r[i] = 0.0;
    }
    if (bad) return kOptFalse;
    cumsum7(r, t);
    return kOptTrue;
}

// The eight v_drive candidates of timeScaling, cc:378-396 (c=1), 408-436 (2), 449-473 (3),
// 485-514 (4), 526-541 (5), 553-567 (6), 579-593 (7), 606-629 (8). v_0/a_0 are already
// mapped to the positive direction (cc:372-375); tr = t_required.
template <int C, int SEM = kSemCpp>
__device__ inline double v_drive_candidate(double am, double jm, const LimPow& P, double q_goal, double q_0, double v_0, double a_0,
                                           double dir, double tr, MatlabCtx& mc)
{
    const double a2 = pw2(a_0), am2 = pw2(am), jm2 = pw2(jm);
    if constexpr (C == 1) {
        return (am * jm * tr / 2 -
                a2 / 4 + a_0 * am / 2 -
                am2 / 2 +
                v_0 * jm / 2 -
                dsqrt(36 * am2 * jm2 * pw2(tr) -
                      36 * a2 * am * jm * tr +
                      72.0 * a_0 * am2 * jm * tr -
                      72.0 * P.am3 * jm * tr +
                      144 * am * dir * jm2 * q_0 -
                      144 * am * dir * jm2 * q_goal +
                      72.0 * am * jm2 * v_0 * tr
                      - 9 * pw4<SEM>(a_0)
                      + 12.0 * pw3<SEM>(a_0) * am
                      + 36 * a2 * am2 +
                      36 * a2 * jm * v_0 -
                      72.0 * a_0 * P.am3 -
                      72.0 * a_0 * am * jm * v_0 +
                      36 * P.am4 -
                      36 * jm2 * pw2(v_0)) / 12) / jm;
    } else if constexpr (C == 2) {
        const double jm3 = P.jm3;
        const double s = a_0 + am;                                       // a_0 + a_max
        const double w = (v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) / am;   // recurring quotient
        const double h = am / (2.0 * jm);
        const double g = (a_0 - am) / (2.0 * jm);
        return -(dir * (q_0 - q_goal) - jm * (
                     pw3<SEM>(s) / (6 * jm3) -
                     P.am3 / (6 * jm3) +
                     (am2 * s) / (2.0 * jm3) +
                     (pw2(s) *
                      (w +
                       h +
                       g)) / (2.0 * jm2)) +
                 a_0 * (pw2(s) / (2.0 * jm2) +
                        am2 / (2.0 * jm2) +
                        (s * (w +
                              h +
                              g)) / jm) -
                 am * (
                     pw2(w - h + g) / 2 +
                     (am * (w - h +
                            g)) / jm) +
                 v_0 * (w +
                        s / jm + h +
                        g)) /
               (h -
                v_0 / am + am * ((w - h +
                                  g) / am + 1.0 / jm) -
                (a2 + 2.0 * a_0 * am +
                 4 * am2 - 2.0 * jm * tr * am +
                 2.0 * jm * v_0) / (2.0 * am * jm) +
                pw2(s) / (2.0 * am * jm) -
                (a_0 * s) / (am * jm));
    } else if constexpr (C == 3) {
        const double c[5] = {
            3.0,
            12.0 * am,
            -24 * am * jm * tr - 12.0 * a2 - 24 * a_0 * am + 12.0 * am2 + 24 * jm * v_0,
            0.0,
            48 * a2 * am * jm * tr -
                96 * dir * jm2 * am * q_0 +
                96 * dir * jm2 * am * q_goal -
                96 * am * jm2 * v_0 * tr +
                12.0 * pw4<SEM>(a_0) +
                16 * pw3<SEM>(a_0) * am -
                24 * a2 * am2 -
                48 * a2 * jm * v_0 +
                48 * am2 * jm * v_0 +
                48 * jm2 * pw2(v_0)};
        const double root2 = root_squared<4, SEM>(c, 3, mc);      // LTPlanner.m:346 root(3)
        return (-2.0 * a2 + 4 * jm * v_0 + root2) / (4 * jm);
    } else if constexpr (C == 4) {
        const double c[5] = {
            12.0,
            24 * am,
            -24 * am * jm * tr + 24 * a2 - 48 * a_0 * am + 24 * am2 - 24 * jm * v_0 + 12.0 * a_0 - 12.0 * am,
            0.0,
            -24 * dir * jm2 * am * q_0 +
                24 * dir * jm2 * am * q_goal +
                9 * pw4<SEM>(a_0) -
                12.0 * pw3<SEM>(a_0) * am -
                24 * a2 * jm * v_0 +
                48 * a_0 * am * jm * v_0 +
                4 * P.am4 -
                24 * am2 * jm * v_0 +
                12.0 * jm2 * pw2(v_0) +
                6 * pw3<SEM>(a_0) +
                6 * a2 * am -
                12.0 * a_0 * am2 -
                12.0 * a_0 * jm * v_0 +
                12.0 * am * jm * v_0 +
                4 * a_0 * am -
                4 * am2};
        const double root2 = root_squared<4, SEM>(c, 3, mc);      // LTPlanner.m:360 root(3)
        return root2 / jm;
    } else if constexpr (C == 5) {
        const double a3 = pw3<SEM>(a_0), jm3 = P.jm3, jm4 = P.jm4, d2 = pw2(dir);
        const double c[6] = {
            (144 * jm * tr + 144 * a_0),
            (-72.0 * jm2 * pw2(tr) - 144 * a_0 * jm * tr + 36 * a2 - 216 * jm * v_0),
            (144 * dir * jm2 * q_0 - 144 * dir * jm2 * q_goal + 48 * a3 - 144 * a_0 * jm * v_0),
            (-144 * dir * jm3 * q_0 * tr + 144 * dir * jm3 * q_goal * tr - 48 * a3 * jm * tr - 144 * a_0 * dir * jm2 * q_0 + 144 * a_0 * dir * jm2 * q_goal + 144 * a_0 * jm2 * v_0 * tr + 6 * pw4<SEM>(a_0) - 72.0 * a2 * jm * v_0 + 216 * jm2 * pw2(v_0)),
            0.0,
            -72.0 * d2 * jm4 * pw2(q_0) + 144 * d2 * jm4 * q_0 * q_goal - 72.0 * d2 * jm4 * pw2(q_goal) - 48 * a3 * dir * jm2 * q_0 + 48 * a3 * dir * jm2 * q_goal + 144 * a_0 * dir * jm3 * q_0 * v_0 - 144 * a_0 * dir * jm3 * q_goal * v_0 + pw6<SEM>(a_0) - 6 * pw4<SEM>(a_0) * jm * v_0 + 36 * a2 * jm2 * pw2(v_0) - 72.0 * jm3 * pw3<SEM>(v_0)};
        const double root2 = root_squared<5, SEM>(c, 2, mc);      // LTPlanner.m:374 root(2)
        return root2 / jm;
    } else if constexpr (C == 6) {
        const double c[5] = {
            3.0,
            -6 * dsqrt(2.0) * am,
            (12.0 * am * jm * tr - 6 * a2 - 12.0 * a_0 * am - 6 * am2 - 12.0 * jm * v_0),
            0.0,
            -12.0 * a2 * am * jm * tr - 24 * dir * jm2 * am * q_0 + 24 * dir * jm2 * am * q_goal - 24 * am * jm2 * v_0 * tr + 3 * pw4<SEM>(a_0) + 4 * pw3<SEM>(a_0) * am + 6 * a2 * am2 + 12.0 * a2 * jm * v_0 + 12.0 * am2 * jm * v_0 + 12.0 * jm2 * pw2(v_0)};
        const double root2 = root_squared<4, SEM>(c, 3, mc);      // LTPlanner.m:388 root(3)
        return -(root2 - a2 - 2.0 * jm * v_0) / (2.0 * jm);
    } else if constexpr (C == 7) {
        const double c[5] = {
            12.0,
            -24 * am,
            (24 * am * jm * tr - 12.0 * a2 - 24 * a_0 * am - 12.0 * am2 - 24 * jm * v_0),
            0.0,
            24 * dir * jm2 * am * q_0 - 24 * dir * jm2 * am * q_goal + 3 * pw4<SEM>(a_0) + 8 * pw3<SEM>(a_0) * am + 6 * a2 * am2 + 12.0 * a2 * jm * v_0 + 24 * a_0 * am * jm * v_0 + 12.0 * am2 * jm * v_0 + 12.0 * jm2 * pw2(v_0)};
        const double root2 = root_squared<4, SEM>(c, 3, mc);      // LTPlanner.m:402 root(3)
        return root2 / jm;
    } else {
        static_assert(C == 8, "case out of range");
        const double a3 = pw3<SEM>(a_0), jm3 = P.jm3, jm4 = P.jm4, d2 = pw2(dir);
        const double c[7] = {
            144.0,
            (-144 * jm * tr + 144 * a_0),
            (72.0 * jm2 * pw2(tr) - 144 * a_0 * jm * tr - 36 * a2 - 216 * jm * v_0),
            (-144 * dir * jm2 * q_0 + 144 * dir * jm2 * q_goal - 48 * a3 - 144 * a_0 * jm * v_0),
            (144 * dir * jm3 * q_0 * tr - 144 * dir * jm3 * q_goal * tr + 48 * a3 * jm * tr - 144 * a_0 * dir * jm2 * q_0 + 144 * a_0 * dir * jm2 * q_goal + 144 * a_0 * jm2 * v_0 * tr + 6 * pw4<SEM>(a_0) + 72.0 * a2 * jm * v_0 + 216 * jm2 * pw2(v_0)),
            0.0,
            72.0 * d2 * jm4 * pw2(q_0) -
                144 * d2 * jm4 * q_0 * q_goal +
                72.0 * d2 * jm4 * pw2(q_goal) +
                48 * a3 * dir * jm2 * q_0 -
                48 * a3 * dir * jm2 * q_goal +
                144 * a_0 * dir * jm3 * q_0 * v_0 -
                144 * a_0 * dir * jm3 * q_goal * v_0 - pw6<SEM>(a_0) -
                6 * pw4<SEM>(a_0) * jm * v_0 -
                36 * a2 * jm2 * pw2(v_0) -
                72.0 * jm3 * pw3<SEM>(v_0)};
        const double root2 = root_squared<6, SEM>(c, 4, mc);      // LTPlanner.m:416 root(4); the C++ notes "WAS root(4) --> Debug this" (cc:628)
        return root2 / jm;
    }
}

// "if (!isnan(v_drive) && v_drive > 0) { optSwitchTimes(...); window test }" — e.g. cc:398-405.
// v_0/a_0 are the direction-mapped values; the reference passes dir*v_0, dir*a_0 on.
// Returns kOptTrue (accepted), kOptFalse (rejected) or, only with FULL == false, kOptDefer.
template <bool FULL, int SEM = kSemCpp>
__device__ inline int try_v_drive(double am, double jm, double vm, const LimPow& P, double t_sample, double q_goal, double q_0, double v_0, double a_0,
                                  double dir, double tr, double v_drive, double (&scaled_t)[7], int& mod, MatlabCtx& mc)
{
    if (!disnan(v_drive) && v_drive > 0.0) {
        double trash;
        const int ok = opt_switch_times<FULL, SEM>(am, jm, vm, P, t_sample, q_goal, q_0, dir * v_0, dir * a_0, v_drive, scaled_t, trash, mod, mc);
        if (ok == kOptDefer) return kOptDefer;
        if (ok == kOptTrue && tr - scaled_t[6] < kTol && tr - scaled_t[6] > -kTol / 10) return kOptTrue;
    }
    return kOptFalse;
}

// ---- timeScaling, all eight candidates in the reference's order (cc:358-645) ----
template <int C, int SEM = kSemCpp>
LTP_DEV bool scaling_case(const JointLimits& L, double t_sample, double qg, double q0, double v0, double a0, double dir,
                          double tr, double& vd, double (&ts)[7], int& mod, MatlabCtx& mc)
{
    vd = v_drive_candidate<C, SEM>(L.a_max, L.j_max, L.pw, qg, q0, v0, a0, dir, tr, mc);
    return try_v_drive<true, SEM>(L.a_max, L.j_max, L.v_max, L.pw, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc) == kOptTrue;
}

// cc:358-645 for one (query, joint): the eight candidates in the reference's order, then the reset
template <int SEM = kSemCpp>
LTP_DEV bool time_scaling_full(const JointLimits& L, double t_sample, double qg, double q0, double v0, double a0, double dir,
                               double tr, double& vd, double (&ts)[7], int& mod, int& which, MatlabCtx& mc)
{
    if (dir < 0.0) { v0 = -v0; a0 = -a0; }   // cc:372-375
    which = 1;
    if (scaling_case<1, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 2;
    if (scaling_case<2, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 3;
    if (scaling_case<3, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 4;
    if (scaling_case<4, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 5;
    if (scaling_case<5, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 6;
    if (scaling_case<6, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 7;
    if (scaling_case<7, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 8;
    if (scaling_case<8, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 0;   // cc:640-644
    mod = 0;
    zero7(ts);
    vd = L.v_max;
    return false;
}

// cc:449-645 for one (query, joint): candidates 3..8 in the reference's order and the reset, for a caller that has already
// evaluated (and seen rejected) candidates 1 and 2. v0 / a0 already mapped to the positive direction (cc:372-375).
template <int SEM = kSemCpp>
LTP_DEV bool time_scaling_tail(const JointLimits& L, double t_sample, double qg, double q0, double v0, double a0, double dir,
                               double tr, double& vd, double (&ts)[7], int& mod, int& which, MatlabCtx& mc)
{
    which = 3;
    if (scaling_case<3, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 4;
    if (scaling_case<4, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 5;
    if (scaling_case<5, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 6;
    if (scaling_case<6, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 7;
    if (scaling_case<7, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 8;
    if (scaling_case<8, SEM>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc)) return true;
    which = 0;   // cc:640-644
    mod = 0;
    zero7(ts);
    vd = L.v_max;
    return false;
}

}  // namespace ltp
