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

namespace ltp {

constexpr double kEps = 4e-3;   // cc:96
constexpr double kTol = 0.1;    // cc:370

struct JointLimits {
    double q_min, q_max, v_max, a_max, j_max;
};

// cc:68-77 for one joint
LTP_DEV bool check_inputs_joint(const JointLimits& L, double q_0, double v_0, double a_0)
{
    if (q_0 < L.q_min || q_0 > L.q_max || dabs(v_0) > L.v_max || dabs(a_0) > L.a_max) return false;
    if (dabs(v_0 + 0.5 * a_0 * dabs(a_0) / L.j_max) > L.v_max) return false;
    return true;
}

// cc:650-701. Writes r[0..2] only.
LTP_DEV void opt_braking(double am, double jm, double t_sample, double v_0, double a_0,
                         double& q, double (&r)[7], double& dir)
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
        r[0] = -a_0 / jm + dsqrt(pw2(a_0) / (2 * pw2(jm)) - v_0 / jm);
        r[2] = r[0] + a_0 / jm;
        r[1] = 0.0;
    }
    q = v_0 * (r[0] + r[1] + r[2]) +
        a_0 * (1.0 / 2.0 * pw2(r[0]) + r[0] * (r[1] + r[2]) + 1.0 / 2.0 * pw2(r[2])) +
        jm * (1.0 / 6.0 * pw3(r[0]) + 1.0 / 2.0 * pw2(r[0]) * (r[1] + r[2]) -
              1.0 / 6.0 * pw3(r[2]) + 1.0 / 2.0 * r[0] * pw2(r[2])) +
        am * (1.0 / 2.0 * pw2(r[1]) + r[1] * r[2]);
    q = dir * q;
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

template <bool FULL>
__device__ inline int opt_switch_times(double am, double jm, double t_sample,
                                       double q_goal, double q_0, double v_0, double a_0, double v_drive,
                                       double (&t)[7], double& dir, int& mod)
{
    double r[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    mod = 0;
    double q_stop = 0.0;
    opt_braking(am, jm, t_sample, v_0, a_0, q_stop, r, dir);
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
        opt_braking(am, jm, t_sample, v_0 - v_drive, a_0, q_brake, r, emp);
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
    double q_part1;
    if (mod == 1) {
        q_part1 = q_brake + v_drive * (r[0] + r[1] + r[2]);
    } else {
        q_part1 = v_0 * (r[0] + r[1] + r[2]) +
                  a_0 * (1.0 / 2.0 * pw2(r[0]) +
                         r[0] * (r[1] + r[2]) +
                         1.0 / 2.0 * pw2(r[2])) +
                  jm * (1.0 / 6.0 * pw3(r[0]) +
                        1.0 / 2.0 * pw2(r[0]) * (r[1] + r[2]) -
                        1.0 / 6.0 * pw3(r[2]) +
                        1.0 / 2.0 * r[0] * pw2(r[2])) +
                  am * (1.0 / 2.0 * pw2(r[1]) + r[1] * r[2]);
    }
    const double q_part2 = jm * (1.0 / 6.0 * pw3(r[6]) +
                                 1.0 / 2.0 * pw2(r[6]) * (r[5] + r[4]) -
                                 1.0 / 6.0 * pw3(r[4]) +
                                 1.0 / 2.0 * r[6] * pw2(r[4])) +
                           am * (1.0 / 2.0 * pw2(r[5]) +
                                 r[5] * r[4]);
    r[3] = ((q_goal - q_0) * dir - q_part1 - q_part2) / v_drive;

    if (r[3] < -kEps) {
        if (mod == 1) {
            zero7(t);
            return kOptFalse;
        }
        const double a2 = pw2(a_0), am2 = pw2(am);
        const double r0_2 = pw2(r[0]), r0_3 = pw3(r[0]), r2_2 = pw2(r[2]), r2_3 = pw3(r[2]);
        const double r4_2 = pw2(r[4]), r4_3 = pw3(r[4]), r6_3 = pw3(r[6]);
        double root = (jm2 * pw4(r[0])) / 2 -
                      (jm2 * pw4(r[2])) / 4 +
                      (jm2 * r2_2 * r4_2) / 2 -
                      (jm2 * pw4(r[4])) / 4 +
                      (jm2 * pw4(r[6])) / 2 +
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
                     2.0 * dsqrt(root) +
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
                        16 * pw3(a_0) - 48 * a_0 * jm * v_0,
                    -3 * pw4(a_0) + 12.0 * a2 * jm * v_0 - 12.0 * jm2 * pw2(v_0)};
                root = smallest_positive_real_root<4>(c);
            }
            r[0] = (2.0 * pw2(root) - 4 * a_0 * root + a2 - 2.0 * v_0 * jm) / (4 * jm * root);
            r[6] = dsqrt(4 * jm2 * pw2(r[0]) +
                         8 * a_0 * jm * r[0] +
                         2.0 * a2 +
                         4 * jm * v_0) / (2.0 * jm);
            r[4] = a_0 / jm + r[0] + r[6];
            r[1] = 0.0;
            r[5] = 0.0;

            if (a_0 + r[0] * jm > am) {
                r[0] = (am - a_0) / jm;
                const double n0_2 = pw2(r[0]);
                r[6] = 1.0 / jm * (am / 2 + dsqrt(
                           9 * am2 + 6 * dsqrt(
                               -12.0 * am * pw3(jm) * pw3(r[0]) +
                               9 * a2 * jm2 * n0_2 -
                               18 * a_0 * am * jm2 * n0_2 +
                               9 * am2 * jm2 * n0_2 +
                               36 * a_0 * jm2 * r[0] * v_0 -
                               72.0 * am * dir * jm2 * q_0 +
                               72.0 * am * dir * jm2 * q_goal -
                               36 * am * jm2 * r[0] * v_0 +
                               3 * pw4(am) +
                               36 * jm2 * pw2(v_0))) / 6.0 - am);
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
                            3 * pw4(a_0) + 8 * pw3(a_0) * am +
                            6 * a2 * am2 -
                            12.0 * a2 * jm * v_0 -
                            24 * a_0 * jm * v_0 * am -
                            12.0 * am2 * jm * v_0 +
                            12.0 * jm2 * pw2(v_0)};
                    root = smallest_positive_real_root<4>(c);
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
template <int C>
__device__ inline double v_drive_candidate(double am, double jm, double q_goal, double q_0, double v_0, double a_0,
                                           double dir, double tr)
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
                      72.0 * pw3(am) * jm * tr +
                      144 * am * dir * jm2 * q_0 -
                      144 * am * dir * jm2 * q_goal +
                      72.0 * am * jm2 * v_0 * tr
                      - 9 * pw4(a_0)
                      + 12.0 * pw3(a_0) * am
                      + 36 * a2 * am2 +
                      36 * a2 * jm * v_0 -
                      72.0 * a_0 * pw3(am) -
                      72.0 * a_0 * am * jm * v_0 +
                      36 * pw4(am) -
                      36 * jm2 * pw2(v_0)) / 12) / jm;
    } else if constexpr (C == 2) {
        const double jm3 = pw3(jm);
        const double s = a_0 + am;                                       // a_0 + a_max
        const double w = (v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) / am;   // recurring quotient
        const double h = am / (2.0 * jm);
        const double g = (a_0 - am) / (2.0 * jm);
        return -(dir * (q_0 - q_goal) - jm * (
                     pw3(s) / (6 * jm3) -
                     pw3(am) / (6 * jm3) +
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
                12.0 * pw4(a_0) +
                16 * pw3(a_0) * am -
                24 * a2 * am2 -
                48 * a2 * jm * v_0 +
                48 * am2 * jm * v_0 +
                48 * jm2 * pw2(v_0)};
        const double root = smallest_positive_real_root<4>(c);
        return (-2.0 * a2 + 4 * jm * v_0 + pw2(root)) / (4 * jm);
    } else if constexpr (C == 4) {
        const double c[5] = {
            12.0,
            24 * am,
            -24 * am * jm * tr + 24 * a2 - 48 * a_0 * am + 24 * am2 - 24 * jm * v_0 + 12.0 * a_0 - 12.0 * am,
            0.0,
            -24 * dir * jm2 * am * q_0 +
                24 * dir * jm2 * am * q_goal +
                9 * pw4(a_0) -
                12.0 * pw3(a_0) * am -
                24 * a2 * jm * v_0 +
                48 * a_0 * am * jm * v_0 +
                4 * pw4(am) -
                24 * am2 * jm * v_0 +
                12.0 * jm2 * pw2(v_0) +
                6 * pw3(a_0) +
                6 * a2 * am -
                12.0 * a_0 * am2 -
                12.0 * a_0 * jm * v_0 +
                12.0 * am * jm * v_0 +
                4 * a_0 * am -
                4 * am2};
        const double root = smallest_positive_real_root<4>(c);
        return pw2(root) / jm;
    } else if constexpr (C == 5) {
        const double a3 = pw3(a_0), jm3 = pw3(jm), jm4 = pw4(jm), d2 = pw2(dir);
        const double c[6] = {
            (144 * jm * tr + 144 * a_0),
            (-72.0 * jm2 * pw2(tr) - 144 * a_0 * jm * tr + 36 * a2 - 216 * jm * v_0),
            (144 * dir * jm2 * q_0 - 144 * dir * jm2 * q_goal + 48 * a3 - 144 * a_0 * jm * v_0),
            (-144 * dir * jm3 * q_0 * tr + 144 * dir * jm3 * q_goal * tr - 48 * a3 * jm * tr - 144 * a_0 * dir * jm2 * q_0 + 144 * a_0 * dir * jm2 * q_goal + 144 * a_0 * jm2 * v_0 * tr + 6 * pw4(a_0) - 72.0 * a2 * jm * v_0 + 216 * jm2 * pw2(v_0)),
            0.0,
            -72.0 * d2 * jm4 * pw2(q_0) + 144 * d2 * jm4 * q_0 * q_goal - 72.0 * d2 * jm4 * pw2(q_goal) - 48 * a3 * dir * jm2 * q_0 + 48 * a3 * dir * jm2 * q_goal + 144 * a_0 * dir * jm3 * q_0 * v_0 - 144 * a_0 * dir * jm3 * q_goal * v_0 + pw6(a_0) - 6 * pw4(a_0) * jm * v_0 + 36 * a2 * jm2 * pw2(v_0) - 72.0 * jm3 * pw3(v_0)};
        const double root = smallest_positive_real_root<5>(c);
        return pw2(root) / jm;
    } else if constexpr (C == 6) {
        const double c[5] = {
            3.0,
            -6 * dsqrt(2.0) * am,
            (12.0 * am * jm * tr - 6 * a2 - 12.0 * a_0 * am - 6 * am2 - 12.0 * jm * v_0),
            0.0,
            -12.0 * a2 * am * jm * tr - 24 * dir * jm2 * am * q_0 + 24 * dir * jm2 * am * q_goal - 24 * am * jm2 * v_0 * tr + 3 * pw4(a_0) + 4 * pw3(a_0) * am + 6 * a2 * am2 + 12.0 * a2 * jm * v_0 + 12.0 * am2 * jm * v_0 + 12.0 * jm2 * pw2(v_0)};
        const double root = smallest_positive_real_root<4>(c);
        return -(pw2(root) - a2 - 2.0 * jm * v_0) / (2.0 * jm);
    } else if constexpr (C == 7) {
        const double c[5] = {
            12.0,
            -24 * am,
            (24 * am * jm * tr - 12.0 * a2 - 24 * a_0 * am - 12.0 * am2 - 24 * jm * v_0),
            0.0,
            24 * dir * jm2 * am * q_0 - 24 * dir * jm2 * am * q_goal + 3 * pw4(a_0) + 8 * pw3(a_0) * am + 6 * a2 * am2 + 12.0 * a2 * jm * v_0 + 24 * a_0 * am * jm * v_0 + 12.0 * am2 * jm * v_0 + 12.0 * jm2 * pw2(v_0)};
        const double root = smallest_positive_real_root<4>(c);
        return pw2(root) / jm;
    } else {
        static_assert(C == 8, "case out of range");
        const double a3 = pw3(a_0), jm3 = pw3(jm), jm4 = pw4(jm), d2 = pw2(dir);
        const double c[7] = {
            144.0,
            (-144 * jm * tr + 144 * a_0),
            (72.0 * jm2 * pw2(tr) - 144 * a_0 * jm * tr - 36 * a2 - 216 * jm * v_0),
            (-144 * dir * jm2 * q_0 + 144 * dir * jm2 * q_goal - 48 * a3 - 144 * a_0 * jm * v_0),
            (144 * dir * jm3 * q_0 * tr - 144 * dir * jm3 * q_goal * tr + 48 * a3 * jm * tr - 144 * a_0 * dir * jm2 * q_0 + 144 * a_0 * dir * jm2 * q_goal + 144 * a_0 * jm2 * v_0 * tr + 6 * pw4(a_0) + 72.0 * a2 * jm * v_0 + 216 * jm2 * pw2(v_0)),
            0.0,
            72.0 * d2 * jm4 * pw2(q_0) -
                144 * d2 * jm4 * q_0 * q_goal +
                72.0 * d2 * jm4 * pw2(q_goal) +
                48 * a3 * dir * jm2 * q_0 -
                48 * a3 * dir * jm2 * q_goal +
                144 * a_0 * dir * jm3 * q_0 * v_0 -
                144 * a_0 * dir * jm3 * q_goal * v_0 - pw6(a_0) -
                6 * pw4(a_0) * jm * v_0 -
                36 * a2 * jm2 * pw2(v_0) -
                72.0 * jm3 * pw3(v_0)};
        const double root = smallest_positive_real_root<6>(c);
        return pw2(root) / jm;
    }
}

// "if (!isnan(v_drive) && v_drive > 0) { optSwitchTimes(...); window test }" — e.g. cc:398-405.
// v_0/a_0 are the direction-mapped values; the reference passes dir*v_0, dir*a_0 on.
// Returns kOptTrue (accepted), kOptFalse (rejected) or, only with FULL == false, kOptDefer.
template <bool FULL>
__device__ inline int try_v_drive(double am, double jm, double t_sample, double q_goal, double q_0, double v_0, double a_0,
                                  double dir, double tr, double v_drive, double (&scaled_t)[7], int& mod)
{
    if (!disnan(v_drive) && v_drive > 0.0) {
        double trash;
        const int ok = opt_switch_times<FULL>(am, jm, t_sample, q_goal, q_0, dir * v_0, dir * a_0, v_drive, scaled_t, trash, mod);
        if (ok == kOptDefer) return kOptDefer;
        if (ok == kOptTrue && tr - scaled_t[6] < kTol && tr - scaled_t[6] > -kTol / 10) return kOptTrue;
    }
    return kOptFalse;
}

// ---- timeScaling, all eight candidates in the reference's order (cc:358-645) ----
template <int C>
LTP_DEV bool scaling_case(const JointLimits& L, double t_sample, double qg, double q0, double v0, double a0, double dir,
                          double tr, double& vd, double (&ts)[7], int& mod)
{
    vd = v_drive_candidate<C>(L.a_max, L.j_max, qg, q0, v0, a0, dir, tr);
    return try_v_drive<true>(L.a_max, L.j_max, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod) == kOptTrue;
}

// cc:358-645 for one (query, joint): the eight candidates in the reference's order, then the reset
LTP_DEV bool time_scaling_full(const JointLimits& L, double t_sample, double qg, double q0, double v0, double a0, double dir,
                               double tr, double& vd, double (&ts)[7], int& mod, int& which)
{
    if (dir < 0.0) { v0 = -v0; a0 = -a0; }   // cc:372-375
    which = 1;
    if (scaling_case<1>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 2;
    if (scaling_case<2>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 3;
    if (scaling_case<3>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 4;
    if (scaling_case<4>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 5;
    if (scaling_case<5>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 6;
    if (scaling_case<6>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 7;
    if (scaling_case<7>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 8;
    if (scaling_case<8>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 0;   // cc:640-644
    mod = 0;
    zero7(ts);
    vd = L.v_max;
    return false;
}

}  // namespace ltp
