// ltp_runs.hpp — how a joint's trajectory is cut into runs: the ONE place that restates the jerk array of getTrajectory
// (cc:735-807) and walks the runs of a joint from its switching-time record. Everything that needs rows or states goes through
// jerk_at() / for_each_run() here and run_coef() / run_eval() of the public header include/ltp_run_tables.hpp — the cooperative
// table build of k_sample (ltp_sampler_lds.hpp), k_build_tables, k_state_at, k_end_limit, k_plan_small — which is what keeps
// their results bit-identical to each other.
#pragma once
#include "ltp_device.hpp"
#include "../../include/ltp_run_tables.hpp"

namespace ltp {

// value of the reference's j_traj[joint][i] after the seven range fills (cc:759-766, last writer wins) and the up
// to eight "+=" fractional corrections (cc:768-807), applied in the reference's order and association (cc:781 and
// cc:798 add two / three terms to the element one after the other). s = sampled switch indices,
// Jp = jerk of the seven phases, corr = the correction terms, all in LDS.
template <int SEM = kSemCpp>
LTP_DEV double jerk_at(const int* s, const double* Jp, const double* corr, int i)
{
    const int s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3], s4 = s[4], s5 = s[5], s6 = s[6];
    double val = 0.0;
    if (s0 > 0 && i < s0) val = Jp[0];
    if (s1 - s0 > 0 && i >= s0 && i < s1) val = Jp[1];
    if (s2 - s1 > 0 && i >= s1 && i < s2) val = Jp[2];
    if (s3 - s2 > 0 && i >= s2 && i < s3) val = Jp[3];
    if (s4 - s3 > 0 && i >= s3 && i < s4) val = Jp[4];
    if (s5 - s4 > 0 && i >= s4 && i < s5) val = Jp[5];
    if (s6 - s5 > 0 && i >= s5 && i < s6) val = Jp[6];
    // LTPlanner.m:558-597 addresses the same elements in a 1-BASED array: every correction lands one sample earlier than in
    // the C++, which kept the index expressions for its 0-based arrays
    constexpr int o = SEM == kSemMatlab ? 1 : 0;
    if (s2 >= s1) {
        if (i == s0 + 1 - o) val = val + corr[0];
        if (s1 > 0 && i == s1 - o) val = val + corr[1];
        if (i == s2 + 1 - o) val = val + corr[2];
    } else {
        if (s1 > 0 && i == s1 - o) val = (val + corr[0]) + corr[3];             // cc:781: j + A + B, left to right
    }
    if (s3 > 0 && i == s3 - o) val = val + corr[4];
    if (s2 - s0 > 0) {
        if (i == s4 + 1 - o) val = val + corr[5];
    } else {
        if (s4 > 0 && i == s4 - o) val = ((val + corr[5]) + corr[0]) + corr[3]; // cc:798: j + A + B + C, left to right
    }
    if (s5 > 0 && i == s5 - o) val = val + corr[7];
    if (i == s6 + 1 - o) val = val + corr[8];
    return val;
}

// One lane walks the runs of one joint in order: the same cut points, jerk_at(), run_coef() and run_eval() as the
// cooperative table build of k_sample, with everything in registers. A kernel whose work per plan is small
// (k_state_at) uses this instead of LDS tables: no block-level build, no per-item latency, 64 independent
// (plan, joint) pairs per wave. (For the envelope consumer the same form is slower than the cooperative kernel,
// 26.8 vs 20.0 ms per 1 M plans: lanes of a wave sit in runs of different lengths.) visit(b, e, rc) is called for every run [b, e) with its
// coefficients and returns true to stop; (a, v, q) hold the state before the run and are advanced to its last sample
// (exactly the value the sampler stores there) after each call that returns false.
// MATLAB's mod(x, y) for y > 0 as LTPlanner.m:531 uses it: x - floor(x./y).*y, except that "if y is not an integer and the
// quotient x./y is within roundoff error of an integer, then n is that integer" (MATLAB documentation), i.e. the result is 0;
// the round-off test is GNU Octave's published rule, as in the test suite's CPU twin. (cc:747 has no such rule.)
LTP_DEV double matlab_mod(double x, double y)
{
    if (y == 0.0) return x;
    const double q = x / y;
    const double n = __builtin_rint(q);
    if (__builtin_rint(y) != y && dabs((q - n) / n) < kDblEps) return 0.0;
    return x - y * dfloor(q);
}

// what the walk reads of a (plan, joint) record: loaded in one place so that a caller can issue the loads ahead of time
struct JointRecord {
    double t[7];            // t_scaled (cc:20, 43-55)
    double dir, v_drive;
    double mod;             // the jerk-profile flag as a double (cc:24)
};
LTP_DEV JointRecord load_joint_record(const Records& rec, long long rj)
{
    JointRecord R;
#pragma unroll
    for (int x = 0; x < 7; ++x) R.t[x] = rec.t_scaled[rj * 7 + x];
    R.dir = rec.dir[rj];
    R.v_drive = rec.v_drive[rj];
    R.mod = (double)rec.mod[rj];
    return R;
}

// LEAN (C++ semantics only): the branch-free form of the walk below (round 6). false: the general form, whose compare masks and
// branches live on the scalar unit — the builder wave of k_sample_walk_* shares its SIMD with five streaming waves, and there the
// scalar work was free while every vector instruction is not (profiles/r06_walk_forms_ab.txt).
template <int SEM = kSemCpp, bool LEAN = true, class Visit>
LTP_DEV void for_each_run_loaded(const JointRecord& R, double j_max, int len, double Ts, double& q, double& v, double& a, Visit&& visit,
                                 bool last_joint = true)
{
    int sw[7];                                                                        // sampled switch indices (cc:751-757)
    double fr[7], frts[7];
#pragma unroll
    for (int x = 0; x < 7; ++x) {
        const double tk = R.t[x];
        fr[x] = SEM == kSemMatlab ? matlab_mod(tk, Ts) : tk - Ts * dfloor(tk / Ts);   // cc:747 / LTPlanner.m:531
        frts[x] = fr[x] / Ts;
        sw[x] = (x & 1) ? (int)dceil(tk / Ts) : (int)dfloor(tk / Ts);
    }
    const double dir = R.dir;
    const double dj = dir * j_max;
    const double vsnap = R.v_drive * dir;                                             // cc:823
    const bool modp = R.mod == 1.0;
    // phase jerks (cc:735-744) and the nine possible correction terms (cc:771-807), as in build_run_tables step (2)
    const double J0 = dj * (modp ? -1.0 : 1.0), J2 = dj * (modp ? 1.0 : -1.0), J4 = dj * -1.0, J6 = dj * 1.0;
    const double Jp[7] = {J0, dj * 0.0, J2, dj * 0.0, J4, dj * 0.0, J6};
    const double d20 = (fr[2] - fr[0]) / Ts;
    const double corr[9] = {frts[0] * J0, (1 - frts[1]) * J2, frts[2] * J2, d20 * J2, (1 - frts[3]) * J4,
                            frts[4] * J4, 0.0, (1 - frts[5]) * J6, frts[6] * J6};
    if constexpr (SEM != kSemMatlab && LEAN) {
        // ---- C++ semantics: the walk below, written without lane branches (round 6) ----
        // One lane's stream is what a builder wave of k_sample_walk_* or a single call waits for, and the general form further down
        // spends ~250 instructions per run on ~45 of arithmetic: compare masks combined on the scalar unit, exec-mask branches around
        // every conditional assignment of jerk_at and around run_coef's modes. Same values, same order of every floating-point
        // operation: the seven range fills become seven unsigned interval tests that select the FACTOR of Jp[k] = dj * {+-1.0, 0.0}
        // (one dword; later ranges override earlier ones as cc:759-766 does), the corrections become ten unconditional additions of
        // either the term or -0.0 (x + -0.0 == x for every x, also for -0.0 and NaN) in the reference's order (cc:768-807), the
        // cut search is a max / select / min per group, and the modes are interval tests.
        int st[7];
        unsigned wid[7];
        st[0] = 0;
        wid[0] = sw[0] > 0 ? (unsigned)sw[0] : 0u;
#pragma unroll
        for (int k = 1; k < 7; ++k) {
            st[k] = sw[k - 1];
            const int d = sw[k] - sw[k - 1];
            wid[k] = d > 0 ? (unsigned)d : 0u;
        }
        constexpr unsigned kOneHi = 0x3ff00000u, kMinusOneHi = 0xbff00000u, kNoneHi = 0x7ff80000u;   // high words of 1.0, -1.0, a NaN
        const unsigned fac[7] = {modp ? kMinusOneHi : kOneHi, 0u, modp ? kOneHi : kMinusOneHi, 0u, kMinusOneHi, 0u, kOneHi};
        // where each correction lands (-1: nowhere, a run never starts below 0) and what is added there, cc:768-807
        const bool o1 = sw[2] >= sw[1], o2 = sw[2] - sw[0] > 0;
        const int p_a = o1 ? sw[0] + 1 : -1;                              // + corr0                                   (cc:771)
        const int p_b = sw[1] > 0 ? sw[1] : -1;                           // + corr1 (cc:774) | + corr0 + corr3        (cc:781)
        const double b1 = o1 ? corr[1] : corr[0], b2 = o1 ? -0.0 : corr[3];
        const int p_c = o1 ? sw[2] + 1 : -1;                              // + corr2                                   (cc:776)
        const int p_d = sw[3] > 0 ? sw[3] : -1;                           // + corr4                                   (cc:787)
        const int p_e = o2 ? sw[4] + 1 : (sw[4] > 0 ? sw[4] : -1);        // + corr5 (cc:793) | + corr5 + corr0 + corr3 (cc:798)
        const double e2 = o2 ? -0.0 : corr[0], e3 = o2 ? -0.0 : corr[3];
        const int p_f = sw[5] > 0 ? sw[5] : -1;                           // + corr7                                   (cc:804)
        const int p_g = sw[6] + 1;                                        // + corr8                                   (cc:807)
        constexpr int cut_lo[7] = {0, 0, 0, -1, 0, 0, 0}, cut_hi[7] = {2, 1, 2, 1, 2, 1, 2};
        int glo[7], ghi[7];
#pragma unroll
        for (int g = 0; g < 7; ++g) { glo[g] = sw[g] + cut_lo[g]; ghi[g] = sw[g] + cut_hi[g]; }
        // cc:813, 822: the constant-velocity samples are s2 + 1 .. s3 - 2 when s3 - s2 > 2
        const int vs0 = sw[2] + 1;
        const unsigned vsw = sw[3] - sw[2] > 2 ? (unsigned)(sw[3] - 1 - vs0) : 0u;
        int b = 0;
        for (int run = 0; run < kMaxSegments && b < len; ++run) {
            int e = len;
#pragma unroll
            for (int g = 0; g < 7; ++g) {
                const int c = glo[g] > b ? glo[g] : b + 1;
                const int cc = c <= ghi[g] ? c : 0x7fffffff;
                e = cc < e ? cc : e;
            }
            const int mode = (b > sw[6] ? kModeTail : 0) | ((unsigned)(b - vs0) < vsw ? kModeVSnap : 0);
            unsigned fh = kNoneHi;
#pragma unroll
            for (int k = 0; k < 7; ++k) fh = (unsigned)(b - st[k]) < wid[k] ? fac[k] : fh;
            double J = dj * __builtin_bit_cast(double, (unsigned long long)fh << 32);
            J = fh == kNoneHi ? 0.0 : J;
            J = J + (b == p_a ? corr[0] : -0.0);
            J = J + (b == p_b ? b1 : -0.0);
            J = J + (b == p_b ? b2 : -0.0);
            J = J + (b == p_c ? corr[2] : -0.0);
            J = J + (b == p_d ? corr[4] : -0.0);
            J = J + (b == p_e ? corr[5] : -0.0);
            J = J + (b == p_e ? e2 : -0.0);
            J = J + (b == p_e ? e3 : -0.0);
            J = J + (b == p_f ? corr[7] : -0.0);
            J = J + (b == p_g ? corr[8] : -0.0);
            const RunCoef rc = run_coef_sel<SEM>(mode, J, a, v, q, vsnap, Ts);
            if (visit(b, e, rc)) return;
            double jj;
            run_eval(rc.c, e - b, q, v, a, jj);
            b = e;
        }
        return;
    }
    // Candidate cut points: every index where the jerk array or a snap rule (cc:815-829) can change — the same set as kCutBase /
    // kCutDelta of the cooperative build: per sampled switch index s_g a few CONSECUTIVE integers s_g + lo_g .. s_g + hi_g. The next
    // cut after b inside group g is therefore max(s_g + lo_g, b + 1) if that is <= s_g + hi_g: three operations per group instead of
    // four per candidate. MATLAB semantics: the corrections sit one sample earlier, the constant-velocity samples are s2 .. s3-2
    // and the tail starts at s6 (LTPlanner.m:616, 620).
    constexpr int cut_lo[7] = {0, 0, 0, -1, 0, 0, 0}, cut_hi[7] = {2, 1, 2, 1, 2, 1, 2};
    constexpr int mcut_lo[7] = {0, -1, 0, -1, -1, -1, 0}, mcut_hi[7] = {1, 0, 1, 0, 1, 0, 1};
    int glo[7], ghi[7];
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        glo[g] = sw[g] + (SEM == kSemMatlab ? mcut_lo[g] : cut_lo[g]);
        ghi[g] = sw[g] + (SEM == kSemMatlab ? mcut_hi[g] : cut_hi[g]);
    }
    const bool phase4 = sw[3] - sw[2] > 2;                                            // cc:813
    int b = 0;
    for (int run = 0; run < kMaxSegments && b < len; ++run) {
        int e = len;                                                                  // next cut point after b
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            const int c = glo[g] > b ? glo[g] : b + 1;
            e = (c <= ghi[g] && c < e) ? c : e;
        }
        int mode = 0;
        if constexpr (SEM == kSemMatlab) {
            if (b >= sw[6]) mode |= last_joint ? kModeTail : (kModeTail | kModeKeepA);
            if (phase4 && b >= sw[2] && b < sw[3] - 1) mode |= kModeVSnap;
        } else {
            if (b > sw[6]) mode |= kModeTail;
            if (phase4 && b >= sw[2] + 1 && b < sw[3] - 1) mode |= kModeVSnap;
        }
        const double J = jerk_at<SEM>(sw, Jp, corr, b);
        const RunCoef rc = run_coef<SEM>(mode, J, a, v, q, vsnap, Ts);
        if (visit(b, e, rc)) return;
        if constexpr (SEM == kSemMatlab) {
            // LTPlanner.m:604-624: a, v, q are cumulative sums over the arrays as they stand — behind the constant-velocity
            // samples (and in the tail) v continues from the UN-snapped sum, the acceleration sum never stops
            const double md = (double)(e - b);
            const double v2 = 0.5 * (Ts * (Ts * J));
            const double v_cum = __builtin_fma(__builtin_fma(v2, md, Ts * a + v2), md, v);   // run_eval's v of an ordinary run
            const double a_cum = __builtin_fma(Ts * J, md, a);
            double vv, aa, jj;
            run_eval(rc.c, e - b, q, vv, aa, jj);
            v = v_cum;
            a = a_cum;
        } else {
            double jj;
            run_eval(rc.c, e - b, q, v, a, jj);
        }
        b = e;
    }
}

template <int SEM = kSemCpp, class Visit>
LTP_DEV void for_each_run(const Limits& lim, const Records& rec, long long rj, int j, int len, double Ts, double& q, double& v,
                          double& a, Visit&& visit, bool last_joint = true)
{
    for_each_run_loaded<SEM, true>(load_joint_record(rec, rj), lim.j_max[j], len, Ts, q, v, a, visit, last_joint);
}

}  // namespace ltp
