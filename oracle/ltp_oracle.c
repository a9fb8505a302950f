/*
 * oracle/ltp_oracle.c — TEST INFRASTRUCTURE ONLY.
 *
 * Scalar, single-threaded CPU restatement (plain C, libm) of the reference
 * planner's hot path, used ONLY as the checker by tests/, by
 * __graft_entry__.smoke() and by bench.py's cpu_baseline leg. Nothing in the
 * product (longtermplanner_amd/, include/) may include, link or call it.
 *
 * Follows, function by function (all paths relative to /root/reference):
 *   src/long_term_planner.cc:7-63     planTrajectory     -> ltpo_plan_trajectory
 *   src/long_term_planner.cc:68-77    checkInputs        -> ltpo_check_inputs
 *   src/long_term_planner.cc:82-353   optSwitchTimes     -> ltpo_opt_switch_times
 *   src/long_term_planner.cc:358-645  timeScaling        -> ltpo_time_scaling
 *   src/long_term_planner.cc:650-701  optBraking         -> ltpo_opt_braking
 *   src/long_term_planner.cc:706-841  getTrajectory      -> ltpo_get_trajectory
 *   include/long_term_planner/long_term_planner.h:54-56  sign<T>
 *   include/long_term_planner/roots.h:22-50              -> companion_roots.inc
 * Every floating-point expression keeps the reference's operand order and
 * association (binary64, no contraction, libm pow/sqrt) so that results agree
 * with the reference's own arithmetic up to libm's last-bit behaviour.
 *
 * PARITY PIN STATUS: the reference translation unit cannot be built in this
 * image (it needs Eigen 3.4, which is absent; writing stand-in headers is not
 * allowed), so this oracle is pinned by the reference's own test tables
 * (tests/src/long_term_planner_tests.cc, tests/src/roots_tests.cc), held as
 * data in tests/golden/ and checked by tests/test_oracle_kat.py, at the
 * tolerances those tests state (1e-3 ... 0.1; 1e-5 for the root finder incl.
 * Eigen's output order). Tighter than that, parity against the TRUE Eigen
 * eigen-solve is unpinned; everything outside the eigen-solve is a direct
 * operation-for-operation restatement.
 *
 * Behaviour the reference leaves undefined and this oracle DEFINES
 * (SURVEY.md App. D), identically to the HIP product:
 *   - sampler writes past traj_len (cc:771,776,793,807) are dropped;
 *   - a plan whose switching times are not finite gets traj_len = 0;
 *   - failed / non-finite eigen-solves select no root (+INFINITY).
 */
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

/* ---------- root finder, instantiated for double and float ---------- */
static int ltpo_schur_iterations = 0;
int ltpo_last_schur_iterations(void) { return ltpo_schur_iterations; }
/* diagnostic: Francis steps of the last solve by size of the active window iu - il + 1 (tools/schur_iters.py) */
static int ltpo_schur_window_steps[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
static int ltpo_schur_iliu_steps[81];   /* [il][iu] */
void ltpo_last_schur_window_steps(int *out9) { int i; for (i = 0; i < 9; i++) out9[i] = ltpo_schur_window_steps[i]; }
void ltpo_last_schur_iliu_steps(int *out81) { int i; for (i = 0; i < 81; i++) out81[i] = ltpo_schur_iliu_steps[i]; }
#define REAL double
#define REAL_MIN DBL_MIN
#define REAL_EPS DBL_EPSILON
#define REAL_SQRT sqrt
#define FN(x) x##_f64
#include "companion_roots.inc"
#undef REAL
#undef REAL_MIN
#undef REAL_EPS
#undef REAL_SQRT
#undef FN

#define REAL float
#define REAL_MIN FLT_MIN
#define REAL_EPS FLT_EPSILON
#define REAL_SQRT sqrtf
#define FN(x) x##_f32
#include "companion_roots.inc"
#undef REAL
#undef REAL_MIN
#undef REAL_EPS
#undef REAL_SQRT
#undef FN

typedef struct {
    int dof;
    double t_sample;
    const double *q_min, *q_max, *v_max, *a_max, *j_max;
    int semantics;   /* 0 = the C++ reference (src/long_term_planner.cc), 1 = the MATLAB original (LTPlanner.m), see below */
} ltpo_planner;

/*
 * MATLAB-semantics mode. With P->semantics == 1 the functions below follow LTPlanner.m wherever
 * it diverges from the C++ translation; every such branch cites the .m lines. What MATLAB does that plain real arithmetic
 * cannot is DEFINED here, identically to the HIP product, and reported through ltpo_matlab_flags():
 *   LTPM_COMPLEX  MATLAB's sqrt of a negative number is a complex value that then flows through the formulas (and a
 *                 filtered polynomial root may carry an imaginary part below eps, LTPlanner.m:247-249). This restatement
 *                 continues with the REAL PART (sqrt -> 0) and applies the safety test |imag(t_rel)| > eps
 *                 (LTPlanner.m:294-297) to the imaginary part where it arose; the plan is delivered and flagged.
 *   LTPM_ERROR    LTPlanner.m would have raised an error: checkInputs (LTPlanner.m:92-103), an index past the filtered
 *                 roots (:250), a vector assigned to t_rel(1) (:275: the filter must leave exactly one root), NaN / Inf
 *                 polynomial coefficients. The query is rejected (status 0, no trajectory).
 * PARITY PIN STATUS of this mode: the three MATLAB unit tables (tests/unittests/ *.m, all seven switching times) and the
 * end-error bounds of the two MATLAB grid tests (tests/gridTestOneJoint.m, gridTestTimeScaling.m), held in
 * tests/golden/reference_kat.json; the eigenvalue ORDER that the positional root picks depend on is pinned against
 * numpy.roots (same LAPACK driver), against MATLAB itself it is unpinned (no MATLAB or Octave in this image).
 */
#define LTPM_COMPLEX 1
#define LTPM_ERROR 2
static _Thread_local int ltpm_flags = 0;
static _Thread_local double ltpm_imag = 0.0;   /* largest imaginary part that entered t_rel in the current optSwitchTimes */
int ltpo_matlab_flags(int clear) { int f = ltpm_flags; if (clear) ltpm_flags = 0; return f; }

#include "matlab_roots.inc"

/* sqrt as LTPlanner.m's arithmetic sees it: MATLAB returns i*sqrt(-x) for x < 0 (C++: NaN) */
static double sem_sqrt(const ltpo_planner *P, double x)
{
    if (P->semantics == 1 && x < 0.0) {
        const double im = sqrt(-x);
        ltpm_flags |= LTPM_COMPLEX;
        if (im > ltpm_imag) ltpm_imag = im;
        return 0.0;
    }
    return sqrt(x);
}

/* exported: eigenvalues in Eigen's order (roots.h:22-34); status as in the .inc */
int ltpo_roots_f64(const double *p, int degree, double *re, double *im)
{
    return companion_eigenvalues_f64(p, degree, re, im);
}
int ltpo_roots_f32(const float *p, int degree, float *re, float *im)
{
    return companion_eigenvalues_f32(p, degree, re, im);
}
/* exported: roots() + getSmallestPositiveNonComplexRoot() as the planner uses them */
double ltpo_smallest_root(const double *p, int degree)
{
    double re[RS_MAXN], im[RS_MAXN];
    companion_eigenvalues_f64(p, degree, re, im);
    return smallest_positive_real_root_f64(re, im, degree);
}

/* ---- test-only build switch -DLTPO_EXACT_POW (oracle/Makefile: libltp_oracle_exactpow.so) ----
 * The reference calls libm's pow(x, 3 | 4 | 6) and pow(x, 1.0 / 2) (cc:125-331, 378-621). glibc's pow is within
 * ~0.52 ulp but not correctly rounded, and dispatches between an FMA and a non-FMA variant by CPU, so the
 * reference's own last bits differ between hosts. The HIP product forms x^3, x^4, x^6 as ONE rounding of the
 * exact product (csrc/ltp_math.hpp: pw3 / pw4 / pw6, error-free products through fma) and pow(x, 0.5) as sqrt.
 * This switch restates exactly that rule in C, so that a comparison "device vs this build" isolates libm's pow
 * as the only source of last-bit differences between the device and the default (libm) oracle
 * (tools/pow_experiment.py, tests/test_gpu_parity.py). The DEFAULT build stays on libm's pow: it is the
 * parity reference. gcc already folds pow(x, 2) to x * x in both builds (no -ffast-math needed). */
#ifdef LTPO_EXACT_POW
static void ltpo_two_prod(double a, double b, double *hi, double *lo) { *hi = a * b; *lo = fma(a, b, -*hi); }
static double ltpo_pow_exact(double x, double y)
{
    double h, l, p, e, e3;
    if (y == 2.0) return x * x;
    if (y == 0.5) return sqrt(x);
    if (y == 3.0) {                                   /* csrc/ltp_math.hpp: pw3 */
        ltpo_two_prod(x, x, &h, &l);
        ltpo_two_prod(h, x, &p, &e);
        return isfinite(p) ? p + (e + l * x) : h * x;
    }
    if (y == 4.0) {                                   /* pw4 */
        ltpo_two_prod(x, x, &h, &l);
        ltpo_two_prod(h, h, &p, &e);
        return isfinite(p) ? p + (e + 2.0 * (h * l)) : h * h;
    }
    if (y == 6.0) {                                   /* pw6 */
        double p3;
        ltpo_two_prod(x, x, &h, &l);
        ltpo_two_prod(h, x, &p3, &e3);
        e3 = e3 + l * x;
        ltpo_two_prod(p3, p3, &p, &e);
        return isfinite(p) ? p + (e + 2.0 * (p3 * e3)) : (h * x) * (h * x);
    }
    return (pow)(x, y);
}
#define pow(x, y) ltpo_pow_exact((x), (y))
int ltpo_exact_pow(void) { return 1; }
#else
int ltpo_exact_pow(void) { return 0; }
#endif

/* exported for tests/: the HOST libm's pow, elementwise — what the product's pow rule LTP_POW_LIBM (csrc/ltp_libm_pow.hpp, a
 * restatement of glibc's pow) is compared with bit for bit. Called through a volatile pointer so that no call is folded. */
void ltpo_libm_pow(long n, const double *x, const double *y, double *out)
{
    double (*volatile f)(double, double) = (pow);
    long i;
    for (i = 0; i < n; i++) out[i] = f(x[i], y[i]);
}

/* optional polynomial log (tests dump the polynomials a run produced) */
static double *g_poly_log = NULL;   /* rows of [degree, p0..p6, root] = 9 doubles */
static long g_poly_cap = 0, g_poly_n = 0;
void ltpo_poly_log_begin(double *buf, long cap_rows) { g_poly_log = buf; g_poly_cap = cap_rows; g_poly_n = 0; }
long ltpo_poly_log_end(void) { long n = g_poly_n; g_poly_log = NULL; g_poly_cap = 0; g_poly_n = 0; return n; }

static double solve_poly(const double *p, int degree)
{
    double r = ltpo_smallest_root(p, degree);
    if (g_poly_log && g_poly_n < g_poly_cap) {
        double *row = g_poly_log + 9 * g_poly_n++;
        int i;
        row[0] = degree;
        for (i = 0; i < 7; i++) row[1 + i] = (i <= degree) ? p[i] : 0.0;
        row[8] = r;
    }
    return r;
}

/*
 * LTPlanner.m:247-250 / 272-275: root = root(abs(imag(root)) < eps); root = root(root >= 0) — MATLAB compares the real
 * parts — then root(1) (site A) or the whole filtered vector assigned to the scalar t_rel(1) (site B: an error unless
 * exactly one root is left). The real part is used; an imaginary part below eps is recorded (sem_sqrt's convention).
 */
static double matlab_filtered_root(const double *c, int deg, double eps, int must_be_single)
{
    double re[MR_MAXN], im[MR_MAXN], pick = NAN, pick_im = 0.0;
    int nr = 0, i, kept = 0;
    const int st = ltpm_roots(c, deg, re, im, &nr);
    if (st != 0) { ltpm_flags |= LTPM_ERROR; return NAN; }
    for (i = 0; i < nr; i++) {
        if (fabs(im[i]) < eps && re[i] >= 0.0) {
            if (kept == 0) { pick = re[i]; pick_im = fabs(im[i]); }
            kept++;
        }
    }
    if (kept == 0 || (must_be_single && kept != 1)) { ltpm_flags |= LTPM_ERROR; return NAN; }
    if (pick_im != 0.0) {
        ltpm_flags |= LTPM_COMPLEX;
        if (pick_im > ltpm_imag) ltpm_imag = pick_im;
    }
    if (g_poly_log && g_poly_n < g_poly_cap) {
        double *row = g_poly_log + 9 * g_poly_n++;
        row[0] = deg;
        for (i = 0; i < 7; i++) row[1 + i] = (i <= deg) ? c[i] : 0.0;
        row[8] = pick;
    }
    return pick;
}

/*
 * The square of the root a timeScaling candidate uses. C++ (cc:467-627): the smallest positive exactly-real root.
 * LTPlanner.m:345-416: root(k) BY POSITION in the output of roots(); v_drive is then tested with ~imag(v_drive), i.e. the
 * square of a complex root must be exactly real, else the candidate is skipped (NaN here has the same effect).
 */
static double root_squared(const ltpo_planner *P, const double *c, int deg, int k)
{
    if (P->semantics == 1) {
        double re[MR_MAXN], im[MR_MAXN], rr, ri;
        int nr = 0, i;
        const int st = ltpm_roots(c, deg, re, im, &nr);
        if (st != 0 || k > nr) { ltpm_flags |= LTPM_ERROR; return NAN; }   /* roots() error / index exceeds the number of elements */
        rr = re[k - 1] * re[k - 1] - im[k - 1] * im[k - 1];
        ri = 2.0 * (re[k - 1] * im[k - 1]);
        if (g_poly_log && g_poly_n < g_poly_cap) {
            double *row = g_poly_log + 9 * g_poly_n++;
            row[0] = deg;
            for (i = 0; i < 7; i++) row[1 + i] = (i <= deg) ? c[i] : 0.0;
            row[8] = im[k - 1] == 0.0 ? re[k - 1] : NAN;
        }
        return ri != 0.0 ? NAN : rr;
    }
    {
        const double root = solve_poly(c, deg);
        return pow(root, 2);
    }
}

/* long_term_planner.h:54-56 */
static int sgn(double v) { return (0.0 < v) - (v < 0.0); }

#define P2(x) pow((x), 2)
#define P3(x) pow((x), 3)
#define P4(x) pow((x), 4)
#define P6(x) pow((x), 6)

/* cc:68-77 */
int ltpo_check_inputs(const ltpo_planner *P, const double *q_0, const double *v_0, const double *a_0)
{
    int i;
    for (i = 0; i < P->dof; i++) {
        /* LTPlanner.m:92-103 has no position limits at all; its three tests raise error() */
        if (P->semantics != 1 && (q_0[i] < P->q_min[i] || q_0[i] > P->q_max[i])) return 0;
        if (fabs(v_0[i]) > P->v_max[i] || fabs(a_0[i]) > P->a_max[i]) return 0;
        if (fabs(v_0[i] + 0.5 * a_0[i] * fabs(a_0[i]) / P->j_max[i]) > P->v_max[i]) return 0;
    }
    return 1;
}

/* diagnostic (tools/pow_experiment.py): which branches the LAST ltpo_opt_switch_times call of this thread went through.
 * 1 optBraking without phase 2 (cc:685-689), 2 modified profile (cc:119-124), 4 phase 2 absent (cc:130-142),
 * 8 phase 6 absent (cc:150-162), 16 no cruise phase: root + pow(root, 1/2) (cc:192-243), 32 quartic site A (cc:245-270),
 * 64 acceleration limit after site A (cc:276-304), 128 quartic site B (cc:306-333), 256 |q_diff| < eps early exit */
static _Thread_local int ltpo_sites = 0;
static _Thread_local double ltpo_site_root = 0.0, ltpo_site_scale = 0.0;   /* site 16: the radicand and its largest term */
int ltpo_last_sites(void) { return ltpo_sites; }
void ltpo_last_site_root(double *out2) { out2[0] = ltpo_site_root; out2[1] = ltpo_site_scale; }

/* cc:650-701. Writes only t_rel[0..2]. */
int ltpo_opt_braking(const ltpo_planner *P, int joint, double v_0, double a_0, double *q, double *t_rel, double *dir)
{
    const double am = P->a_max[joint], jm = P->j_max[joint];
    if (v_0 * a_0 > 0) {
        *dir = -sgn(v_0);
    } else {
        if (fabs(v_0) > 1.0 / 2.0 * P2(a_0) / jm) *dir = -sgn(v_0);
        else *dir = -sgn(a_0);
    }
    if (*dir < 0) {
        a_0 = -a_0;
        v_0 = -v_0;
    }
    t_rel[0] = (am - a_0) / jm;
    t_rel[2] = am / jm;
    t_rel[1] = (-v_0 - 1.0 / 2.0 * t_rel[0] * a_0) / am - 1.0 / 2.0 * (t_rel[0] + t_rel[2]);
    if (t_rel[1] < -P->t_sample) {
        ltpo_sites |= 1;
        t_rel[0] = -a_0 / jm + sem_sqrt(P, P2(a_0) / (2 * P2(jm)) - v_0 / jm);   /* LTPlanner.m:476: complex for a negative argument */
        t_rel[2] = t_rel[0] + a_0 / jm;
        t_rel[1] = 0;
    }
    *q = v_0 * (t_rel[0] + t_rel[1] + t_rel[2]) +
         a_0 * (1.0 / 2.0 * P2(t_rel[0]) + t_rel[0] * (t_rel[1] + t_rel[2]) + 1.0 / 2.0 * P2(t_rel[2])) +
         jm * (1.0 / 6.0 * P3(t_rel[0]) + 1.0 / 2.0 * P2(t_rel[0]) * (t_rel[1] + t_rel[2]) -
               1.0 / 6.0 * P3(t_rel[2]) + 1.0 / 2.0 * t_rel[0] * P2(t_rel[2])) +
         am * (1.0 / 2.0 * P2(t_rel[1]) + t_rel[1] * t_rel[2]);
    *q = *dir * *q;
    return 1;
}

static void cumsum7(const double *r, double *t)
{
    int i;
    double s = r[0];
    t[0] = s;
    for (i = 1; i < 7; i++) { s = s + r[i]; t[i] = s; }
}
static void zero7(double *t) { int i; for (i = 0; i < 7; i++) t[i] = 0.0; }

/* cc:82-353. Returns the reference's bool; `t` is written only where the reference writes it. */
int ltpo_opt_switch_times(const ltpo_planner *P, int joint, double q_goal, double q_0, double v_0, double a_0,
                          double v_drive, double *t, double *dir, char *mod)
{
    const double am = P->a_max[joint], jm = P->j_max[joint];
    const double eps = 4e-3;
    double r[7] = {0, 0, 0, 0, 0, 0, 0};
    double q_stop = 0, q_diff, q_brake = 0.0, emp, q_part1 = 0, q_part2;
    int i;
    const int matlab = P->semantics == 1;
    *mod = 0;
    if (matlab) {
        /* LTPlanner.m:131 -> :92-103: checkInputs inside optSwitchTimes, error() on violation */
        ltpm_imag = 0.0;
        if (fabs(v_0) > P->v_max[joint] || fabs(a_0) > P->a_max[joint] ||
            fabs(v_0 + 1.0 / 2.0 * a_0 * fabs(a_0) / jm) > P->v_max[joint]) {
            ltpm_flags |= LTPM_ERROR;
            zero7(t);
            return 0;
        }
    }

    ltpo_sites = 0;
    ltpo_opt_braking(P, joint, v_0, a_0, &q_stop, r, dir);
    q_diff = q_goal - (q_0 + q_stop);
    if (fabs(q_diff) < eps) {
        ltpo_sites |= 256;
        cumsum7(r, t);
        return 1;
    }
    *dir = sgn(q_diff);
    if (*dir < 0) {
        v_0 = -v_0;
        a_0 = -a_0;
    }
    if (v_0 + 0.5 * a_0 * fabs(a_0) / jm > v_drive) {
        *mod = 1;
        ltpo_sites |= 2;
        ltpo_opt_braking(P, joint, v_0 - v_drive, a_0, &q_brake, r, &emp);
    } else {
        r[0] = (am - a_0) / jm;
        r[2] = am / jm;
        r[1] = (v_drive - v_0 - 0.5 * r[0] * a_0) / am - 0.5 * (r[0] + r[2]);
        if (r[1] < -eps) {
            double root = jm * (v_drive - v_0) + 0.5 * P2(a_0);
            ltpo_sites |= 4;
            if (root > 0) {
                r[2] = sqrt(root) / jm;
                r[0] = r[2] - a_0 / jm;
                r[1] = 0;
            } else {
                zero7(t);
                return 1;
            }
        }
    }
    r[4] = am / jm;
    r[6] = r[4];
    r[5] = v_drive / am - 1.0 / 2.0 * (r[4] + r[6]);
    if (r[5] < -eps) {
        double root = v_drive / jm;
        ltpo_sites |= 8;
        if (root > 0) {
            r[4] = sqrt(root);
            r[6] = r[4];
            r[5] = 0;
        } else {
            zero7(t);
            return 1;
        }
    }
    if (*mod == 1) {
        q_part1 = q_brake + v_drive * (r[0] + r[1] + r[2]);
    } else {
        q_part1 = v_0 * (r[0] + r[1] + r[2]) +
                  a_0 * (1.0 / 2.0 * P2(r[0]) +
                         r[0] * (r[1] + r[2]) +
                         1.0 / 2.0 * P2(r[2])) +
                  jm * (1.0 / 6.0 * P3(r[0]) +
                        1.0 / 2.0 * P2(r[0]) * (r[1] + r[2]) -
                        1.0 / 6.0 * P3(r[2]) +
                        1.0 / 2.0 * r[0] * P2(r[2])) +
                  am * (1.0 / 2.0 * P2(r[1]) + r[1] * r[2]);
    }
    q_part2 = jm * (1.0 / 6.0 * P3(r[6]) +
                    1.0 / 2.0 * P2(r[6]) * (r[5] + r[4]) -
                    1.0 / 6.0 * P3(r[4]) +
                    1.0 / 2.0 * r[6] * P2(r[4])) +
              am * (1.0 / 2.0 * P2(r[5]) +
                    r[5] * r[4]);
    r[3] = ((q_goal - q_0) * *dir - q_part1 - q_part2) / v_drive;

    if (r[3] < -eps) {
        double root;
        if (*mod == 1) {
            zero7(t);
            return matlab ? 1 : 0;   /* LTPlanner.m:222-227 returns the zeros like any other result; cc:195-200 returns false */
        }
        root = (P2(jm) * P4(r[0])) / 2 -
               (P2(jm) * P4(r[2])) / 4 +
               (P2(jm) * P2(r[2]) * P2(r[4])) / 2 -
               (P2(jm) * P4(r[4])) / 4 +
               (P2(jm) * P4(r[6])) / 2 +
               2.0 * jm * a_0 * P3(r[0]) -
               (2.0 * jm * am * P3(r[0])) / 3 -
               2.0 * jm * am * r[0] * P2(r[2]) +
               (2.0 * jm * am * P3(r[2])) / 3 +
               (2.0 * jm * am * P3(r[4])) / 3 -
               2.0 * jm * am * P2(r[4]) * r[6] -
               (2.0 * jm * am * P3(r[6])) / 3 +
               2.0 * jm * v_0 * P2(r[0]) +
               2.0 * P2(a_0) * P2(r[0]) -
               2.0 * a_0 * am * P2(r[0]) -
               2.0 * a_0 * am * P2(r[2]) +
               4 * a_0 * v_0 * r[0] +
               2.0 * P2(am) * P2(r[2]) +
               2.0 * P2(am) * P2(r[4]) -
               4 * am * v_0 * r[0] +
               4 * *dir * (q_goal - q_0) * am +
               2.0 * P2(v_0);
        ltpo_sites |= 16;
        ltpo_site_root = root;
        ltpo_site_scale = fabs(4 * *dir * (q_goal - q_0) * am) + fabs((P2(jm) * P4(r[0])) / 2) + fabs(2.0 * P2(am) * P2(r[2]));
        if (root > 0) {
            r[5] = -(4 * am * r[4] -
                     2.0 * pow(root, (1.0 / 2)) +
                     jm * P2(r[2]) -
                     jm * P2(r[4]) +
                     2.0 * jm * P2(r[6])) / (4 * am);
            r[1] = (-v_0 - a_0 * r[0] -
                    1.0 / 2.0 * jm * P2(r[0]) +
                    1.0 / 2.0 * jm * P2(r[2]) +
                    1.0 / 2.0 * jm * P2(r[6]) -
                    1.0 / 2.0 * jm * P2(r[4])) / am
                   - r[2] + r[5] + r[4];
            r[3] = 0;
        } else {
            zero7(t);
            return 1;
        }

        if (r[5] < -eps || r[1] < -eps) {
            double c[5];
            ltpo_sites |= 32;
            c[0] = 12;
            c[1] = 0;
            c[2] = -24 * P2(a_0) + 48 * jm * v_0;
            c[3] = 48 * *dir * P2(jm) * q_0 -
                   48 * *dir * P2(jm) * q_goal +
                   16 * P3(a_0) - 48 * a_0 * jm * v_0;
            c[4] = -3 * P4(a_0) + 12.0 * P2(a_0) * jm * v_0 - 12.0 * P2(jm) * P2(v_0);
            root = matlab ? matlab_filtered_root(c, 4, eps, 0) : solve_poly(c, 4);   /* LTPlanner.m:247-250: first root that passes the filter */
            if (matlab && (ltpm_flags & LTPM_ERROR)) { zero7(t); return 0; }
            r[0] = (2.0 * P2(root) - 4 * a_0 * root + P2(a_0) - 2.0 * v_0 * jm) / (4 * jm * root);
            r[6] = sem_sqrt(P, 4 * P2(jm) * P2(r[0]) +
                        8 * a_0 * jm * r[0] +
                        2.0 * P2(a_0) +
                        4 * jm * v_0) / (2.0 * jm);
            r[4] = a_0 / jm + r[0] + r[6];
            r[1] = 0;
            r[5] = 0;

            if (a_0 + r[0] * jm > am) {
                ltpo_sites |= 64;
                r[0] = (am - a_0) / jm;
                r[6] = 1.0 / jm * (am / 2 + sem_sqrt(P,
                           9 * P2(am) + 6 * sem_sqrt(P,
                               -12.0 * am * P3(jm) * P3(r[0]) +
                               9 * P2(a_0) * P2(jm) * P2(r[0]) -
                               18 * a_0 * am * P2(jm) * P2(r[0]) +
                               9 * P2(am) * P2(jm) * P2(r[0]) +
                               36 * a_0 * P2(jm) * r[0] * v_0 -
                               72.0 * am * *dir * P2(jm) * q_0 +
                               72.0 * am * *dir * P2(jm) * q_goal -
                               36 * am * P2(jm) * r[0] * v_0 +
                               3 * P4(am) +
                               36 * P2(jm) * P2(v_0))) / 6.0 - am);
                r[4] = r[6] + am / jm;
                r[1] = -(-jm * P2(r[4]) -
                         2.0 * jm * r[4] * r[6] +
                         jm * P2(r[6]) + a_0 * r[0] +
                         am * r[0] +
                         2.0 * am * r[4] +
                         2.0 * am * r[6] +
                         2.0 * v_0) / (2.0 * am);
                r[5] = 0;
            }

            if (r[6] * jm > am) {
                ltpo_sites |= 128;
                r[6] = am / jm;
                c[0] = 12;
                c[1] = -24 * am;
                c[2] = -12.0 * P2(a_0) + 12.0 * P2(am) + 24 * jm * v_0;
                c[3] = 0;
                c[4] = 24 * *dir * P2(jm) * q_0 * am -
                       24 * *dir * P2(jm) * q_goal * am +
                       3 * P4(a_0) + 8 * P3(a_0) * am +
                       6 * P2(a_0) * P2(am) -
                       12.0 * P2(a_0) * jm * v_0 -
                       24 * a_0 * jm * v_0 * am -
                       12.0 * P2(am) * jm * v_0 +
                       12.0 * P2(jm) * P2(v_0);
                root = matlab ? matlab_filtered_root(c, 4, eps, 1) : solve_poly(c, 4);   /* LTPlanner.m:272-275: the filter must leave exactly one root */
                if (matlab && (ltpm_flags & LTPM_ERROR)) { zero7(t); return 0; }
                r[0] = (root - a_0 - am) / jm;
                r[4] = (a_0 + am) / jm + r[0];
                r[5] = (P2(jm) * P2(r[0]) +
                        2.0 * P2(jm) * r[0] * r[4] -
                        P2(jm) * P2(r[4]) +
                        2.0 * a_0 * jm * r[0] +
                        2.0 * a_0 * jm * r[4] -
                        P2(am) +
                        2.0 * jm * v_0) / (2.0 * jm * am);
                r[1] = 0;
            }
            r[2] = 0;
            r[3] = 0;
        }
    }
    if (matlab) {
        /* LTPlanner.m:288-303: any(t_rel < -eps) or any(|imag(t_rel)| > eps) zeroes t_rel (no failure); then
         * t_rel = max(0, real(t_rel)), which also turns NaN into 0 (MATLAB's max ignores NaN) */
        int bad = 0;
        for (i = 0; i < 7; i++) if (r[i] < -eps) bad = 1;
        if (ltpm_imag > eps) bad = 1;
        for (i = 0; i < 7; i++) r[i] = (bad || !(r[i] > 0.0)) ? 0.0 : r[i];
        cumsum7(r, t);
        return 1;
    }
    /* cc:340-348 (the std::cerr diagnostic is not reproduced) */
    for (i = 0; i < 7; i++) {
        if (r[i] < -eps) {
            return 0;
        } else if (r[i] < 0.0 && r[i] >= -eps) {
            r[i] = 0.0;
        }
    }
    cumsum7(r, t);
    return 1;
}

/* the acceptance test repeated after each candidate, e.g. cc:398-405 */
static int try_v_drive(const ltpo_planner *P, int joint, double q_goal, double q_0, double v_0, double a_0,
                       double dir, double t_required, double v_drive, double *scaled_t, char *mod)
{
    const double tol = 0.1;
    if (!isnan(v_drive) && v_drive > 0) {
        double trash;
        int ok = ltpo_opt_switch_times(P, joint, q_goal, q_0, dir * v_0, dir * a_0, v_drive, scaled_t, &trash, mod);
        if (ok && t_required - scaled_t[6] < tol && t_required - scaled_t[6] > -tol / 10) return 1;
    }
    return 0;
}

/* cc:358-645. *case_out (may be NULL) receives 1..8 for the accepted case, 0 for none. */
int ltpo_time_scaling_ex(const ltpo_planner *P, int joint, double q_goal, double q_0, double v_0, double a_0,
                         double dir, double t_required, double *scaled_t, double *v_drive, char *mod, int *case_out)
{
    const double am = P->a_max[joint], jm = P->j_max[joint];
    const double tr = t_required;
    double c[7], root2;
    int dummy;
    if (!case_out) case_out = &dummy;
    if (dir < 0) {
        v_0 = -v_0;
        a_0 = -a_0;
    }
    /* c1: standard profile, phases 2 and 6 exist (cc:378-405) */
    *v_drive = (am * jm * tr / 2 -
                P2(a_0) / 4 + a_0 * am / 2 -
                P2(am) / 2 +
                v_0 * jm / 2 -
                sqrt(36 * P2(am) * P2(jm) * P2(tr) -
                     36 * P2(a_0) * am * jm * tr +
                     72.0 * a_0 * P2(am) * jm * tr -
                     72.0 * P3(am) * jm * tr +
                     144 * am * dir * P2(jm) * q_0 -
                     144 * am * dir * P2(jm) * q_goal +
                     72.0 * am * P2(jm) * v_0 * tr
                     - 9 * P4(a_0)
                     + 12.0 * P3(a_0) * am
                     + 36 * P2(a_0) * P2(am) +
                     36 * P2(a_0) * jm * v_0 -
                     72.0 * a_0 * P3(am) -
                     72.0 * a_0 * am * jm * v_0 +
                     36 * P4(am) -
                     36 * P2(jm) * P2(v_0)) / 12) / jm;
    *case_out = 1;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* c2: modified profile, phases 2 and 6 exist (cc:408-446) */
    *v_drive = -(dir * (q_0 - q_goal) - jm * (
                   pow(a_0 + am, 3) / (6 * P3(jm)) -
                   P3(am) / (6 * P3(jm)) +
                   (P2(am) * (a_0 + am)) / (2.0 * P3(jm)) +
                   (pow(a_0 + am, 2) *
                    ((v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) / am +
                     am / (2.0 * jm) +
                     (a_0 - am) / (2.0 * jm))) / (2.0 * P2(jm))) +
                 a_0 * (pow(a_0 + am, 2) / (2.0 * P2(jm)) +
                        P2(am) / (2.0 * P2(jm)) +
                        ((a_0 + am) * ((v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) / am +
                                       am / (2.0 * jm) +
                                       (a_0 - am) / (2.0 * jm))) / jm) -
                 am * (
                   pow((v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) / am - am / (2.0 * jm) + (a_0 - am) / (2.0 * jm), 2) / 2 +
                   (am * ((v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) / am - am / (2.0 * jm) +
                          (a_0 - am) / (2.0 * jm))) / jm) +
                 v_0 * ((v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) / am +
                        (a_0 + am) / jm + am / (2.0 * jm) +
                        (a_0 - am) / (2.0 * jm))) /
               (am / (2.0 * jm) -
                v_0 / am + am * (((v_0 + (a_0 * (a_0 - am)) / (2.0 * jm)) /
                                  am - am / (2.0 * jm) +
                                  (a_0 - am) / (2.0 * jm)) / am + 1.0 / jm) -
                (P2(a_0) + 2.0 * a_0 * am +
                 4 * P2(am) - 2.0 * jm * tr * am +
                 2.0 * jm * v_0) / (2.0 * am * jm) +
                pow(a_0 + am, 2) / (2.0 * am * jm) -
                (a_0 * (a_0 + am)) / (am * jm));
    *case_out = 2;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* c3: standard profile, phase 2 does not exist (cc:449-482) */
    c[0] = 3;
    c[1] = 12.0 * am;
    c[2] = -24 * am * jm * tr - 12.0 * P2(a_0) - 24 * a_0 * am + 12.0 * P2(am) + 24 * jm * v_0;
    c[3] = 0;
    c[4] = 48 * P2(a_0) * am * jm * tr -
           96 * dir * P2(jm) * am * q_0 +
           96 * dir * P2(jm) * am * q_goal -
           96 * am * P2(jm) * v_0 * tr +
           12.0 * P4(a_0) +
           16 * P3(a_0) * am -
           24 * P2(a_0) * P2(am) -
           48 * P2(a_0) * jm * v_0 +
           48 * P2(am) * jm * v_0 +
           48 * P2(jm) * P2(v_0);
    root2 = root_squared(P, c, 4, 3);      /* LTPlanner.m:346 root(3) */
    *v_drive = (-2.0 * P2(a_0) + 4 * jm * v_0 + root2) / (4 * jm);
    *case_out = 3;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* c4: standard profile, phase 6 does not exist (cc:485-523) */
    c[0] = 12;
    c[1] = 24 * am;
    c[2] = -24 * am * jm * tr + 24 * P2(a_0) - 48 * a_0 * am + 24 * P2(am) - 24 * jm * v_0 + 12.0 * a_0 - 12.0 * am;
    c[3] = 0;
    c[4] = -24 * dir * P2(jm) * am * q_0 +
           24 * dir * P2(jm) * am * q_goal +
           9 * P4(a_0) -
           12.0 * P3(a_0) * am -
           24 * P2(a_0) * jm * v_0 +
           48 * a_0 * am * jm * v_0 +
           4 * P4(am) -
           24 * P2(am) * jm * v_0 +
           12.0 * P2(jm) * P2(v_0) +
           6 * P3(a_0) +
           6 * P2(a_0) * am -
           12.0 * a_0 * P2(am) -
           12.0 * a_0 * jm * v_0 +
           12.0 * am * jm * v_0 +
           4 * a_0 * am -
           4 * P2(am);
    root2 = root_squared(P, c, 4, 3);      /* LTPlanner.m:360 root(3) */
    *v_drive = root2 / jm;
    *case_out = 4;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* c5: standard profile, phases 2 and 6 do not exist (cc:526-550), degree 5 */
    c[0] = (144 * jm * tr + 144 * a_0);
    c[1] = (-72.0 * P2(jm) * P2(tr) - 144 * a_0 * jm * tr + 36 * P2(a_0) - 216 * jm * v_0);
    c[2] = (144 * dir * P2(jm) * q_0 - 144 * dir * P2(jm) * q_goal + 48 * P3(a_0) - 144 * a_0 * jm * v_0);
    c[3] = (-144 * dir * P3(jm) * q_0 * tr + 144 * dir * P3(jm) * q_goal * tr - 48 * P3(a_0) * jm * tr - 144 * a_0 * dir * P2(jm) * q_0 + 144 * a_0 * dir * P2(jm) * q_goal + 144 * a_0 * P2(jm) * v_0 * tr + 6 * P4(a_0) - 72.0 * P2(a_0) * jm * v_0 + 216 * P2(jm) * P2(v_0));
    c[4] = 0;
    c[5] = -72.0 * P2(dir) * P4(jm) * P2(q_0) + 144 * P2(dir) * P4(jm) * q_0 * q_goal - 72.0 * P2(dir) * P4(jm) * P2(q_goal) - 48 * P3(a_0) * dir * P2(jm) * q_0 + 48 * P3(a_0) * dir * P2(jm) * q_goal + 144 * a_0 * dir * P3(jm) * q_0 * v_0 - 144 * a_0 * dir * P3(jm) * q_goal * v_0 + P6(a_0) - 6 * P4(a_0) * jm * v_0 + 36 * P2(a_0) * P2(jm) * P2(v_0) - 72.0 * P3(jm) * P3(v_0);
    root2 = root_squared(P, c, 5, 2);      /* LTPlanner.m:374 root(2) */
    *v_drive = root2 / jm;
    *case_out = 5;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* c6: modified profile, phase 2 does not exist (cc:553-576) */
    c[0] = 3;
    c[1] = -6 * sqrt(2) * am;
    c[2] = (12.0 * am * jm * tr - 6 * P2(a_0) - 12.0 * a_0 * am - 6 * P2(am) - 12.0 * jm * v_0);
    c[3] = 0;
    c[4] = -12.0 * P2(a_0) * am * jm * tr - 24 * dir * P2(jm) * am * q_0 + 24 * dir * P2(jm) * am * q_goal - 24 * am * P2(jm) * v_0 * tr + 3 * P4(a_0) + 4 * P3(a_0) * am + 6 * P2(a_0) * P2(am) + 12.0 * P2(a_0) * jm * v_0 + 12.0 * P2(am) * jm * v_0 + 12.0 * P2(jm) * P2(v_0);
    root2 = root_squared(P, c, 4, 3);      /* LTPlanner.m:388 root(3) */
    *v_drive = -(root2 - P2(a_0) - 2.0 * jm * v_0) / (2.0 * jm);
    *case_out = 6;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* c7: modified profile, phase 6 does not exist (cc:579-603) */
    c[0] = 12;
    c[1] = -24 * am;
    c[2] = (24 * am * jm * tr - 12.0 * P2(a_0) - 24 * a_0 * am - 12.0 * P2(am) - 24 * jm * v_0);
    c[3] = 0;
    c[4] = 24 * dir * P2(jm) * am * q_0 - 24 * dir * P2(jm) * am * q_goal + 3 * P4(a_0) + 8 * P3(a_0) * am + 6 * P2(a_0) * P2(am) + 12.0 * P2(a_0) * jm * v_0 + 24 * a_0 * am * jm * v_0 + 12.0 * P2(am) * jm * v_0 + 12.0 * P2(jm) * P2(v_0);
    root2 = root_squared(P, c, 4, 3);      /* LTPlanner.m:402 root(3) */
    *v_drive = root2 / jm;
    *case_out = 7;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* c8: modified profile, phases 2 and 6 do not exist (cc:606-638), degree 6 */
    c[0] = 144;
    c[1] = (-144 * jm * tr + 144 * a_0);
    c[2] = (72.0 * P2(jm) * P2(tr) - 144 * a_0 * jm * tr - 36 * P2(a_0) - 216 * jm * v_0);
    c[3] = (-144 * dir * P2(jm) * q_0 + 144 * dir * P2(jm) * q_goal - 48 * P3(a_0) - 144 * a_0 * jm * v_0);
    c[4] = (144 * dir * P3(jm) * q_0 * tr - 144 * dir * P3(jm) * q_goal * tr + 48 * P3(a_0) * jm * tr - 144 * a_0 * dir * P2(jm) * q_0 + 144 * a_0 * dir * P2(jm) * q_goal + 144 * a_0 * P2(jm) * v_0 * tr + 6 * P4(a_0) + 72.0 * P2(a_0) * jm * v_0 + 216 * P2(jm) * P2(v_0));
    c[5] = 0;
    c[6] = 72.0 * P2(dir) * P4(jm) * P2(q_0) -
           144 * P2(dir) * P4(jm) * q_0 * q_goal +
           72.0 * P2(dir) * P4(jm) * P2(q_goal) +
           48 * P3(a_0) * dir * P2(jm) * q_0 -
           48 * P3(a_0) * dir * P2(jm) * q_goal +
           144 * a_0 * dir * P3(jm) * q_0 * v_0 -
           144 * a_0 * dir * P3(jm) * q_goal * v_0 - P6(a_0) -
           6 * P4(a_0) * jm * v_0 -
           36 * P2(a_0) * P2(jm) * P2(v_0) -
           72.0 * P3(jm) * P3(v_0);
    root2 = root_squared(P, c, 6, 4);      /* LTPlanner.m:416 root(4); the C++ notes "WAS root(4) --> Debug this" (cc:628) */
    *v_drive = root2 / jm;
    *case_out = 8;
    if (try_v_drive(P, joint, q_goal, q_0, v_0, a_0, dir, tr, *v_drive, scaled_t, mod)) return 1;

    /* cc:640-644 */
    *mod = 0;
    zero7(scaled_t);
    *v_drive = P->v_max[joint];
    *case_out = 0;
    return 0;
}

int ltpo_time_scaling(const ltpo_planner *P, int joint, double q_goal, double q_0, double v_0, double a_0,
                      double dir, double t_required, double *scaled_t, double *v_drive, char *mod)
{
    return ltpo_time_scaling_ex(P, joint, q_goal, q_0, v_0, a_0, dir, t_required, scaled_t, v_drive, mod, NULL);
}

/* cc:716-719; DEFINED: any non-finite t[i][6] -> 0 (the reference casts NaN to int, UB) */
int ltpo_traj_len(const ltpo_planner *P, const double *t /* [dof][7] */)
{
    int i, len = 0;
    for (i = 0; i < P->dof; i++) {
        int k, l;
        for (k = 0; k < 7; k++) if (!isfinite(t[7 * i + k])) return 0;
        if (!(ceil(t[7 * i + 6] / P->t_sample) + 1.0 < 2147483647.0)) return 0; /* DEFINED: length does not fit an int */
        l = (int)ceil(t[7 * i + 6] / P->t_sample) + 1;
        if (l > len) len = l;
    }
    return len;
}


/*
 * MATLAB's mod(x, y) for y > 0 as LTPlanner.m:531 uses it: x - floor(x./y).*y, except that "if y is not an integer and the
 * quotient x./y is within roundoff error of an integer, then n is that integer" (MATLAB documentation of mod), i.e. the
 * result is 0. The round-off test is restated with GNU Octave's published rule (|q - round(q)| / |round(q)| < eps): MATLAB's
 * own constant is not documented. The C++ translation computes t - Ts*floor(t/Ts) without that rule (cc:747).
 */
static double matlab_mod(double x, double y)
{
    double q, n;
    if (y == 0.0) return x;
    q = x / y;
    n = nearbyint(q);
    if (nearbyint(y) != y && fabs((q - n) / n) < DBL_EPSILON) return 0.0;
    {
        volatile double tmp = y * floor(q);
        return x - tmp;
    }
}

/*
 * LTPlanner.m:486-625 getTrajectories, 1-based there, 0-based here (index k of MATLAB is k-1). Differences to cc:706-841:
 *   - the fractional corrections land one sample EARLIER (the C++ kept MATLAB's index expressions in 0-based arrays);
 *   - a = Ts*cumsum(j) + a_0, v = Ts*cumsum(a) + v_0, q = Ts*cumsum(v) + q_0 (:604, :610, :624): the sums run over the
 *     arrays as they stand, so v continues from the un-snapped sum after the constant-velocity samples;
 *   - the constant-velocity samples are S3+1 .. S4-1 (1-based, :616) and the tail starts at S7+1 (:607, :620), one sample
 *     earlier than in the C++;
 *   - :607 "a_traj(joint, sampled_t(joint,7)+1:end) = 0" stands AFTER the joint loop: only the LAST joint's acceleration
 *     tail is zeroed (reproduced as is);
 *   - sample fractions come from mod() (:531).
 * j is returned as well (LTPlanner.m does not return it).
 */
static void matlab_get_trajectories(const ltpo_planner *P, const double *t, const double *dir, const char *mod,
                                    const double *q_0, const double *v_0, const double *a_0, const double *v_drive,
                                    int len, double *q, double *v, double *a, double *j)
{
    const double Ts = P->t_sample;
    int joint;
    memset(j, 0, sizeof(double) * (size_t)P->dof * len);
#define JADDM(idx1, val) do { int ix_ = (idx1) - 1; if (ix_ >= 0 && ix_ < len) jt[ix_] = jt[ix_] + (val); } while (0)
    for (joint = 0; joint < P->dof; joint++) {
        const double *tj = t + 7 * joint;
        double *jt = j + (size_t)joint * len, *at = a + (size_t)joint * len;
        double *vt = v + (size_t)joint * len, *qt = q + (size_t)joint * len;
        double fr[7], jp[7], cs;
        int s[7], prof[7], k, i, const_v;
        if (mod[joint]) { int m[7] = {-1, 0, 1, 0, -1, 0, 1}; memcpy(prof, m, sizeof m); }
        else { int m[7] = {1, 0, -1, 0, -1, 0, 1}; memcpy(prof, m, sizeof m); }
        for (k = 0; k < 7; k++) jp[k] = dir[joint] * P->j_max[joint] * prof[k];
        for (k = 0; k < 7; k++) fr[k] = matlab_mod(tj[k], Ts);
        s[0] = (int)floor(tj[0] / Ts);
        s[1] = (int)ceil(tj[1] / Ts);
        s[2] = (int)floor(tj[2] / Ts);
        s[3] = (int)ceil(tj[3] / Ts);
        s[4] = (int)floor(tj[4] / Ts);
        s[5] = (int)ceil(tj[5] / Ts);
        s[6] = (int)floor(tj[6] / Ts);
        /* :544-551, 1-based 1..S1 and S(k-1)+1..S(k) */
        if (s[0] > 0) for (i = 0; i < s[0] && i < len; i++) jt[i] = jp[0];
        for (k = 1; k < 7; k++)
            if (s[k] - s[k - 1] > 0) for (i = s[k - 1]; i < s[k] && i < len; i++) jt[i] = jp[k];
        /* :554-597 */
        if (s[2] >= s[1]) {
            JADDM(s[0] + 1, fr[0] / Ts * jp[0]);
            if (s[1] > 0) JADDM(s[1], (1 - fr[1] / Ts) * jp[2]);
            JADDM(s[2] + 1, fr[2] / Ts * jp[2]);
        } else {
            if (s[1] > 0) { JADDM(s[1], fr[0] / Ts * jp[0]); JADDM(s[1], (fr[2] - fr[0]) / Ts * jp[2]); }
        }
        if (s[3] > 0) JADDM(s[3], (1 - fr[3] / Ts) * jp[4]);
        if (s[2] - s[0] > 0) {
            JADDM(s[4] + 1, fr[4] / Ts * jp[4]);
        } else {
            if (s[4] > 0) { JADDM(s[4], fr[4] / Ts * jp[4]); JADDM(s[4], fr[0] / Ts * jp[0]); JADDM(s[4], (fr[2] - fr[0]) / Ts * jp[2]); }
        }
        if (s[5] > 0) JADDM(s[5], (1 - fr[5] / Ts) * jp[6]);
        JADDM(s[6] + 1, fr[6] / Ts * jp[6]);
        const_v = s[3] - s[2] > 2;                                            /* :600-602 */
        /* :604 */
        cs = 0.0;
        for (i = 0; i < len; i++) { cs = cs + jt[i]; at[i] = Ts * cs + a_0[joint]; }
        /* :607, outside the joint loop in LTPlanner.m: `joint` is DoF there */
        if (joint == P->dof - 1) for (i = s[6]; i < len; i++) if (i >= 0) at[i] = 0.0;
        /* :610 */
        cs = 0.0;
        for (i = 0; i < len; i++) { cs = cs + at[i]; vt[i] = Ts * cs + v_0[joint]; }
        /* :615-617, 1-based S3+1 .. S4-1 */
        if (const_v) for (i = s[2]; i <= s[3] - 2 && i < len; i++) if (i >= 0) vt[i] = v_drive[joint] * dir[joint];
        /* :620 */
        for (i = s[6]; i < len; i++) if (i >= 0) vt[i] = 0.0;
        /* :624 */
        cs = 0.0;
        for (i = 0; i < len; i++) { cs = cs + vt[i]; qt[i] = Ts * cs + q_0[joint]; }
    }
#undef JADDM
}

/*
 * cc:706-841. Caller provides q,v,a,j as [dof][len] row-major, len = ltpo_traj_len().
 * Writes with an index >= len are dropped (DEFINED; reference UB).
 */
void ltpo_get_trajectory(const ltpo_planner *P, const double *t /* [dof][7] */, const double *dir, const char *mod,
                         const double *q_0, const double *v_0, const double *a_0, const double *v_drive,
                         int len, double *q, double *v, double *a, double *j)
{
    const double Ts = P->t_sample;
    int joint;
    if (len <= 0) return;
    if (P->semantics == 1) {
        matlab_get_trajectories(P, t, dir, mod, q_0, v_0, a_0, v_drive, len, q, v, a, j);
        return;
    }
    memset(q, 0, sizeof(double) * (size_t)P->dof * len);
    memset(v, 0, sizeof(double) * (size_t)P->dof * len);
    memset(a, 0, sizeof(double) * (size_t)P->dof * len);
    memset(j, 0, sizeof(double) * (size_t)P->dof * len);
#define JADD(idx, val) do { int ix_ = (idx); if (ix_ >= 0 && ix_ < len) jt[ix_] = jt[ix_] + (val); } while (0)
    for (joint = 0; joint < P->dof; joint++) {
        const double *tj = t + 7 * joint;
        double *jt = j + (size_t)joint * len, *at = a + (size_t)joint * len;
        double *vt = v + (size_t)joint * len, *qt = q + (size_t)joint * len;
        double fr[7], jp[7];
        int s[7], prof[7], k, i, phase4;
        if (mod[joint] == 1) { int m[7] = {-1, 0, 1, 0, -1, 0, 1}; memcpy(prof, m, sizeof m); }
        else { int m[7] = {1, 0, -1, 0, -1, 0, 1}; memcpy(prof, m, sizeof m); }
        for (k = 0; k < 7; k++) jp[k] = dir[joint] * P->j_max[joint] * prof[k];
        for (k = 0; k < 7; k++) fr[k] = tj[k] - Ts * floor(tj[k] / Ts);
        s[0] = (int)floor(tj[0] / Ts);
        s[1] = (int)ceil(tj[1] / Ts);
        s[2] = (int)floor(tj[2] / Ts);
        s[3] = (int)ceil(tj[3] / Ts);
        s[4] = (int)floor(tj[4] / Ts);
        s[5] = (int)ceil(tj[5] / Ts);
        s[6] = (int)floor(tj[6] / Ts);
        /* fills (cc:759-766); clipped to len (DEFINED) */
        if (s[0] > 0) for (i = 0; i < s[0] && i < len; i++) jt[i] = jp[0];
        for (k = 1; k < 7; k++)
            if (s[k] - s[k - 1] > 0) for (i = s[k - 1]; i < s[k] && i < len; i++) jt[i] = jp[k];
        /* fractional corrections (cc:768-807) */
        if (s[2] >= s[1]) {
            JADD(s[0] + 1, fr[0] / Ts * jp[0]);
            if (s[1] > 0) JADD(s[1], (1 - fr[1] / Ts) * jp[2]);
            JADD(s[2] + 1, fr[2] / Ts * jp[2]);
        } else {
            /* cc:781: j = j + A + B, i.e. (j + A) + B: the terms are added one by one, left to right */
            if (s[1] > 0) { JADD(s[1], fr[0] / Ts * jp[0]); JADD(s[1], (fr[2] - fr[0]) / Ts * jp[2]); }
        }
        if (s[3] > 0) JADD(s[3], (1 - fr[3] / Ts) * jp[4]);
        if (s[2] - s[0] > 0) {
            JADD(s[4] + 1, fr[4] / Ts * jp[4]);
        } else {
            /* cc:798: ((j + A) + B) + C */
            if (s[4] > 0) { JADD(s[4], fr[4] / Ts * jp[4]); JADD(s[4], fr[0] / Ts * jp[0]); JADD(s[4], (fr[2] - fr[0]) / Ts * jp[2]); }
        }
        if (s[5] > 0) JADD(s[5], (1 - fr[5] / Ts) * jp[6]);
        JADD(s[6] + 1, fr[6] / Ts * jp[6]);
        /* integration (cc:810-831) */
        at[0] = a_0[joint] + Ts * jt[0];
        vt[0] = v_0[joint] + Ts * at[0];
        qt[0] = q_0[joint] + Ts * vt[0];
        phase4 = s[3] - s[2] > 2;
        for (i = 1; i < len; i++) {
            if (i <= s[6]) at[i] = at[i - 1] + Ts * jt[i];
            else at[i] = 0.0;
            if (phase4 && i >= s[2] + 1 && i < s[3] - 1) vt[i] = v_drive[joint] * dir[joint];
            else if (i <= s[6]) vt[i] = vt[i - 1] + Ts * at[i];
            else vt[i] = 0.0;
            qt[i] = qt[i - 1] + Ts * vt[i];
        }
    }
#undef JADD
}

/*
 * cc:7-63 up to (not including) the sampler: stages 1-3 + the fallback copy.
 * Returns 1 if the reference would reach getTrajectory, 0 if it returns false before.
 * Output arrays: t_opt,t_scaled [dof][7]; dirv,v_drive [dof]; mod [dof].
 */
int ltpo_plan_switch_times(const ltpo_planner *P, const double *q_goal, const double *q_0, const double *v_0,
                           const double *a_0, double *t_opt, double *t_scaled, double *dirv, char *mod,
                           double *v_drive, double *t_required, int *slowest)
{
    int i, k, D = P->dof;
    *t_required = -1;
    *slowest = -1;
    for (i = 0; i < D; i++) {
        for (k = 0; k < 7; k++) { t_opt[7 * i + k] = 0; t_scaled[7 * i + k] = 0; }
        dirv[i] = 0; mod[i] = 0; v_drive[i] = P->v_max[i];
    }
    if (P->semantics == 1) ltpm_flags &= ~LTPM_ERROR;
    if (!ltpo_check_inputs(P, q_0, v_0, a_0)) { if (P->semantics == 1) ltpm_flags |= LTPM_ERROR; return 0; }
    for (i = 0; i < D; i++) {
        if (!ltpo_opt_switch_times(P, i, q_goal[i], q_0[i], v_0[i], a_0[i], P->v_max[i], t_opt + 7 * i, dirv + i, mod + i)) return 0;
        if (P->semantics == 1) mod[i] = 0;   /* LTPlanner.m:64 discards optSwitchTimes' third output: mod_jerk_profile stays false */
    }
    for (i = 0; i < D; i++) {
        if (t_opt[7 * i + 6] > *t_required) {
            *t_required = t_opt[7 * i + 6];
            *slowest = i;
        }
    }
    if (*slowest == -1) return 0;
    for (i = 0; i < D; i++) {
        if (i == *slowest) continue;
        ltpo_time_scaling(P, i, q_goal[i], q_0[i], v_0[i], a_0[i], dirv[i], *t_required, t_scaled + 7 * i, v_drive + i, mod + i);
    }
    if (P->semantics == 1 && (ltpm_flags & LTPM_ERROR)) return 0;   /* an error() inside timeScaling ends LTPlanner.m's trajectory() */
    for (i = 0; i < D; i++) {
        if (P->semantics == 1) {
            /* LTPlanner.m:82 ~any(t_scaled(joint,:)): every entry exactly zero (NaN counts as non-zero) */
            int any = 0;
            for (k = 0; k < 7; k++) if (t_scaled[7 * i + k] != 0.0) any = 1;
            if (!any) for (k = 0; k < 7; k++) t_scaled[7 * i + k] = t_opt[7 * i + k];
            continue;
        }
        double mx = t_scaled[7 * i];
        for (k = 1; k < 7; k++) if (mx < t_scaled[7 * i + k]) mx = t_scaled[7 * i + k]; /* std::max_element */
        if (mx <= 0.0) for (k = 0; k < 7; k++) t_scaled[7 * i + k] = t_opt[7 * i + k];
    }
    return 1;
}

/*
 * Full cc:7-63 for one query with the reference's allocation pattern (four
 * zero-initialised [dof][len] arrays per call, cc:725-728). Status:
 *   0 = returned false before sampling (traj untouched), 1 = true,
 *   2 = false from the end-limit check cc:59-61 (traj filled).
 * If keep != NULL the four arrays are handed to the caller (free with ltpo_free),
 * else they are released; *len_out gets traj_len; *checksum (optional) the sum of
 * the last sample of q over joints.
 */
static int plan_trajectory_impl(const ltpo_planner *P, const double *q_goal, const double *q_0, const double *v_0,
                                const double *a_0, double *t_opt, double *t_scaled, double *dirv, char *mod,
                                double *v_drive, double *t_required, int *slowest, int *len_out,
                                double **keep /* [4] or NULL */, double *checksum,
                                double **reuse /* [4] or NULL: caller-owned arrays grown on demand */, size_t *reuse_cap)
{
    int D = P->dof, len, i, status = 1;
    double *q, *v, *a, *j;
    *len_out = 0;
    if (!ltpo_plan_switch_times(P, q_goal, q_0, v_0, a_0, t_opt, t_scaled, dirv, mod, v_drive, t_required, slowest)) return 0;
    len = ltpo_traj_len(P, t_scaled);
    *len_out = len;
    if (len <= 0) return 0; /* DEFINED: non-finite switching times */
    if (reuse) {
        size_t need = (size_t)D * len;
        if (need > *reuse_cap) {
            for (i = 0; i < 4; i++) { free(reuse[i]); reuse[i] = (double *)malloc(sizeof(double) * need); }
            *reuse_cap = need;
        }
        q = reuse[0]; v = reuse[1]; a = reuse[2]; j = reuse[3];
    } else {
        q = (double *)malloc(sizeof(double) * (size_t)D * len);
        v = (double *)malloc(sizeof(double) * (size_t)D * len);
        a = (double *)malloc(sizeof(double) * (size_t)D * len);
        j = (double *)malloc(sizeof(double) * (size_t)D * len);
    }
    ltpo_get_trajectory(P, t_scaled, dirv, mod, q_0, v_0, a_0, v_drive, len, q, v, a, j);
    for (i = 0; i < D && P->semantics != 1; i++) {   /* LTPlanner.m has no position limits: no end-limit check */
        double qe = q[(size_t)i * len + len - 1];
        if (qe < P->q_min[i] || qe > P->q_max[i]) { status = 2; break; }
    }
    if (checksum) {
        double s = 0;
        for (i = 0; i < D; i++) s += q[(size_t)i * len + len - 1];
        *checksum = s;
    }
    if (reuse) return status;
    if (keep) { keep[0] = q; keep[1] = v; keep[2] = a; keep[3] = j; }
    else { free(q); free(v); free(a); free(j); }
    return status;
}

int ltpo_plan_trajectory(const ltpo_planner *P, const double *q_goal, const double *q_0, const double *v_0,
                         const double *a_0, double *t_opt, double *t_scaled, double *dirv, char *mod,
                         double *v_drive, double *t_required, int *slowest, int *len_out,
                         double **keep /* [4] or NULL */, double *checksum)
{
    return plan_trajectory_impl(P, q_goal, q_0, v_0, a_0, t_opt, t_scaled, dirv, mod, v_drive, t_required, slowest, len_out,
                                keep, checksum, NULL, NULL);
}

void ltpo_free(void *p) { free(p); }

/*
 * Batch driver used by tests and by bench.py's cpu_baseline: runs queries
 * [first, first+count) of row-major [n][dof] inputs through ltpo_plan_trajectory
 * (or stages 1-3 only when sample == 0; sample == 2 samples into four arrays that
 * are allocated once and reused, the "flat preallocated" CPU variant of BASELINE.md
 * §3, instead of the reference's per-plan allocation). Per-query outputs are optional (NULL).
 * Returns the number of queries with status 1.
 */
long ltpo_plan_batch(const ltpo_planner *P, long first, long count, const double *q_goal, const double *q_0,
                     const double *v_0, const double *a_0, int sample,
                     double *t_opt, double *t_scaled, double *dirv, char *mod, double *v_drive,
                     double *t_required, int *slowest, int *traj_len, int *status, double *checksum)
{
    int D = P->dof;
    long n_ok = 0, p;
    double *b_topt = (double *)malloc(sizeof(double) * 7 * D), *b_tsc = (double *)malloc(sizeof(double) * 7 * D);
    double *b_dir = (double *)malloc(sizeof(double) * D), *b_vd = (double *)malloc(sizeof(double) * D);
    char *b_mod = (char *)malloc(D);
    double *reuse[4] = {NULL, NULL, NULL, NULL};
    size_t reuse_cap = 0;
    for (p = first; p < first + count; p++) {
        double treq, cs = 0;
        int slow, len = 0, st;
        if (P->semantics == 1) ltpm_flags = 0;
        double *o_topt = t_opt ? t_opt + (size_t)p * 7 * D : b_topt;
        double *o_tsc = t_scaled ? t_scaled + (size_t)p * 7 * D : b_tsc;
        double *o_dir = dirv ? dirv + (size_t)p * D : b_dir;
        double *o_vd = v_drive ? v_drive + (size_t)p * D : b_vd;
        char *o_mod = mod ? mod + (size_t)p * D : b_mod;
        if (sample) {
            st = plan_trajectory_impl(P, q_goal + (size_t)p * D, q_0 + (size_t)p * D, v_0 + (size_t)p * D, a_0 + (size_t)p * D,
                                      o_topt, o_tsc, o_dir, o_mod, o_vd, &treq, &slow, &len, NULL, &cs,
                                      sample == 2 ? reuse : NULL, &reuse_cap);
        } else {
            st = ltpo_plan_switch_times(P, q_goal + (size_t)p * D, q_0 + (size_t)p * D, v_0 + (size_t)p * D, a_0 + (size_t)p * D,
                                        o_topt, o_tsc, o_dir, o_mod, o_vd, &treq, &slow);
            if (st) len = ltpo_traj_len(P, o_tsc);
        }
        if (t_required) t_required[p] = treq;
        if (slowest) slowest[p] = slow;
        if (traj_len) traj_len[p] = len;
        /* MATLAB semantics: bits 4 / 5 of the status word carry LTPM_COMPLEX / LTPM_ERROR of this plan */
        if (status) status[p] = P->semantics == 1 ? (st | (ltpm_flags << 4)) : st;
        if (checksum) checksum[p] = cs;
        if (st == 1) n_ok++;
    }
    free(b_topt); free(b_tsc); free(b_dir); free(b_vd); free(b_mod);
    free(reuse[0]); free(reuse[1]); free(reuse[2]); free(reuse[3]);
    return n_ok;
}

/*
 * Dense comparison for the parity soak (tools/dense_soak.py): plans [first, first+count) of row-major [n][dof] inputs are
 * planned and sampled by THIS oracle (full cc:7-63) and compared sample by sample with a device's packed rows
 * (`packed`: host copy of the device tile that starts at element `base`; plan p at offsets[p] - base, laid out
 * [q,v,a,j][dof][row stride = traj_len rounded up to 32]). Per plan: maxd[4] = max |d| over q, v, a, j (a NaN on one
 * side only counts as +inf) and flag bits: 1 = one side planned the query and the other rejected it,
 * 2 = trajectory lengths differ (nothing compared), 4 = end-limit verdicts (cc:59-61) differ, 8 = compared and the JERK
 * rows are not bit-identical (informational; q, v, a come from a different but equivalent summation order on the device).
 * 16 = visited (set for every plan of the call: the caller verifies that each plan was visited, i.e. no range was skipped).
 * dev_status: the device's LTP_STATUS_* word (bits 1|2|4|16|64 = rejected before sampling, 8 = end limit).
 * Returns the number of values compared.
 */
long long ltpo_compare_dense(const ltpo_planner *P, long first, long count, const double *q_goal, const double *q_0,
                             const double *v_0, const double *a_0, const double *packed, const unsigned long long *offsets,
                             unsigned long long base, const int *dev_len, const int *dev_status, double *maxd, int *flag)
{
    int D = P->dof, x, i, k;
    long p;
    long long compared = 0;
    double *t_opt = (double *)malloc(sizeof(double) * 7 * D), *t_sc = (double *)malloc(sizeof(double) * 7 * D);
    double *dirv = (double *)malloc(sizeof(double) * D), *vd = (double *)malloc(sizeof(double) * D);
    char *mod = (char *)malloc(D);
    double *reuse[4] = {NULL, NULL, NULL, NULL};
    size_t reuse_cap = 0;
    for (p = first; p < first + count; p++) {
        double treq;
        int slow, len = 0, st;
        double *md = maxd + 4 * (size_t)(p - first);
        int *fl = flag + (p - first);
        const int dev_rejected = (dev_status[p] & (1 | 2 | 4 | 16 | 64)) != 0;
        md[0] = md[1] = md[2] = md[3] = 0.0;
        *fl = 16;                                            /* visited: the caller checks that every plan was, exactly once per call */
        st = plan_trajectory_impl(P, q_goal + (size_t)p * D, q_0 + (size_t)p * D, v_0 + (size_t)p * D, a_0 + (size_t)p * D,
                                  t_opt, t_sc, dirv, mod, vd, &treq, &slow, &len, NULL, NULL, reuse, &reuse_cap);
        if ((st == 0) != dev_rejected) { *fl |= 1; continue; }
        if (st == 0) continue;
        if (len != dev_len[p]) { *fl |= 2; continue; }
        if ((st == 2) != ((dev_status[p] & 8) != 0)) *fl |= 4;
        {
            const size_t stride = ((size_t)len + 31) / 32 * 32;
            const double *blk = packed + (offsets[p] - base);
            for (x = 0; x < 4; x++) {
                double m = 0.0;
                for (i = 0; i < D; i++) {
                    const double *dev = blk + ((size_t)x * D + i) * stride;
                    const double *ref = reuse[x] + (size_t)i * len;
                    for (k = 0; k < len; k++) {
                        double d;
                        if (dev[k] == ref[k]) d = 0.0;                                  /* includes inf == inf */
                        else if (dev[k] != dev[k] && ref[k] != ref[k]) d = 0.0;         /* NaN on both sides */
                        else { d = fabs(dev[k] - ref[k]); if (!(d == d)) d = INFINITY; }
                        if (d > m) m = d;
                        if (d != 0.0 && x == 3) *fl |= 8;
                    }
                }
                md[x] = m;
            }
            compared += 4ll * D * len;
        }
    }
    free(t_opt); free(t_sc); free(dirv); free(vd); free(mod);
    free(reuse[0]); free(reuse[1]); free(reuse[2]); free(reuse[3]);
    return compared;
}
