/*
 * oracle/kat_grid.c — TEST INFRASTRUCTURE ONLY.
 *
 * Restates the PROCEDURE of the reference's two grid integration tests so the
 * oracle can be pinned by them (the loops are the reference tests' own; the
 * functions under test are the oracle's):
 *   tests/src/long_term_planner_tests.cc:264-323  gridTestOneJoint
 *   tests/src/long_term_planner_tests.cc:325-407  GridTimeScalingTest
 * Each returns the number of failed expectations and reports the number of
 * expectations checked plus the worst goal error seen.
 */
#include <math.h>
#include <stdlib.h>

typedef struct {
    int dof;
    double t_sample;
    const double *q_min, *q_max, *v_max, *a_max, *j_max;
    int semantics;   /* 0 = C++ reference, 1 = LTPlanner.m (see ltp_oracle.c) */
} ltpo_planner;

int ltpo_opt_switch_times(const ltpo_planner *, int, double, double, double, double, double, double *, double *, char *);
int ltpo_time_scaling_ex(const ltpo_planner *, int, double, double, double, double, double, double, double *, double *, char *, int *);
int ltpo_traj_len(const ltpo_planner *, const double *);
void ltpo_get_trajectory(const ltpo_planner *, const double *, const double *, const char *, const double *, const double *,
                         const double *, const double *, int, double *, double *, double *, double *);

static double dmin(double a, double b) { return b < a ? b : a; }
static double dmax(double a, double b) { return a < b ? b : a; }

/* final (q, v, a) of the sampled 1-DoF trajectory */
static int final_state(const ltpo_planner *P, const double *t, double dir, char mod, double q_0, double v_0, double a_0,
                       double v_drive, double *qe, double *ve, double *ae)
{
    int len = ltpo_traj_len(P, t);
    double *buf;
    if (len <= 0) return 0;
    buf = (double *)malloc(sizeof(double) * 4 * (size_t)len);
    ltpo_get_trajectory(P, t, &dir, &mod, &q_0, &v_0, &a_0, &v_drive, len, buf, buf + len, buf + 2 * len, buf + 3 * len);
    *qe = buf[len - 1];
    *ve = buf[2 * len - 1];
    *ae = buf[3 * len - 1];
    free(buf);
    return len;
}

long ltpo_kat_grid_one_joint(long *n_checks, double *worst_err)
{
    const double eps = 1e-6, tol = 0.02, step = 0.1;
    const double q_min[1] = {-3.1}, q_max[1] = {3.1}, v_max[1] = {1.0}, a_max[1] = {2.0}, j_max[1] = {15.0};
    const double q_0 = 0.5;
    ltpo_planner P = {1, 0.004, q_min, q_max, v_max, a_max, j_max, 0};
    long fails = 0;
    int i, j, k;
    *n_checks = 0;
    *worst_err = 0;
    for (i = (int)q_min[0] / step; i <= (int)q_max[0] / step; i++) {
        double q_goal = i * step;
        for (j = (int)-v_max[0] / step; j < (int)v_max[0] / step; j++) {
            double v_0 = j * step;
            double a_lb, a_ub;
            if (v_0 >= 0) {
                a_lb = -(a_max[0] - eps);
                a_ub = dmin(a_max[0] - eps, sqrt(2 * j_max[0] * (v_max[0] - v_0)));
            } else {
                a_lb = dmax(-(a_max[0] - eps), -sqrt(2 * j_max[0] * (v_max[0] - fabs(v_0))));
                a_ub = a_max[0];
            }
            for (k = (int)a_lb / step; k < (int)a_ub / step; k++) {
                double a_0 = k * step - eps;
                double t[7] = {0, 0, 0, 0, 0, 0, 0}, dir, qe = 0, ve, ae, err;
                char mod;
                int ok = ltpo_opt_switch_times(&P, 0, q_goal, q_0, v_0, a_0, v_max[0], t, &dir, &mod);
                (*n_checks)++;
                if (!ok) { fails++; }
                if (!final_state(&P, t, dir, mod, q_0, v_0, a_0, v_max[0], &qe, &ve, &ae)) { fails++; continue; }
                err = fabs(qe - q_goal);
                (*n_checks)++;
                if (!(err <= tol)) fails++;
                if (err > *worst_err) *worst_err = err;
            }
        }
    }
    return fails;
}

/* case_hist[0..8]: how often timeScaling ended in "none" (0) or case 1..8; mod_hist[0..1]: the calls accepted in case 1 or 2
 * split by the mod_jerk_profile flag they return (standard / modified profile); *sum_err / *n_err: sum and count of the goal
 * errors |q_end - q_goal| (README.md:128-136 quotes their mean and maximum) */
long ltpo_kat_grid_time_scaling_stats(long *n_checks, double *worst_err, long *case_hist, long *mod_hist, double *sum_err, long *n_err)
{
    const double eps = 1e-6, tol_q = 0.02, tol_t = 0.1, step = 0.1;
    const double q_min[1] = {-6}, q_max[1] = {7}, v_max[1] = {1.0}, a_max[1] = {2.0}, j_max[1] = {15.0};
    const double q_0 = 0.5;
    const double incr[6] = {0.05, 0.1, 0.2, 0.5, 1.0, 2.0};
    ltpo_planner P = {1, 0.004, q_min, q_max, v_max, a_max, j_max, 0};
    long fails = 0;
    int i, j, k, l, c;
    *n_checks = 0;
    *worst_err = 0;
    for (c = 0; c < 9; c++) case_hist[c] = 0;
    mod_hist[0] = mod_hist[1] = 0;
    *sum_err = 0;
    *n_err = 0;
    for (i = (int)q_min[0] / step; i <= (int)q_max[0] / step; i++) {
        double q_goal = i * step;
        for (j = (int)-v_max[0] / step; j < (int)v_max[0] / step; j++) {
            double v_0 = j * step;
            double a_lb, a_ub, a_range;
            int n_steps;
            v_0 = (j > 0) ? v_0 - eps : v_0 + eps;
            if (v_0 >= 0) {
                a_lb = -(a_max[0] - eps);
                a_ub = dmin(a_max[0] - eps, sqrt(2 * j_max[0] * (v_max[0] - v_0)));
            } else {
                a_lb = dmax(-(a_max[0] - eps), -sqrt(2 * j_max[0] * (v_max[0] - fabs(v_0))));
                a_ub = a_max[0];
            }
            a_range = a_ub - a_lb;
            n_steps = (int)floor(a_range / step);
            for (k = 0; k < n_steps; k++) {
                double a_0 = a_lb + k * step;
                double t_ltp[7] = {0, 0, 0, 0, 0, 0, 0}, dir;
                char mod;
                int ok = ltpo_opt_switch_times(&P, 0, q_goal, q_0, v_0, a_0, v_max[0], t_ltp, &dir, &mod);
                (*n_checks)++;
                if (!ok) { fails++; continue; }
                for (l = 0; l < 6; l++) {
                    double t_scaled[7] = {0, 0, 0, 0, 0, 0, 0}, v_drive, qe = 0, ve = 0, ae = 0, err;
                    char mod2 = 0;
                    int okts, cs = 0, m;
                    if (t_ltp[6] < tol_q) break;
                    okts = ltpo_time_scaling_ex(&P, 0, q_goal, q_0, v_0, a_0, dir, t_ltp[6] + incr[l], t_scaled, &v_drive, &mod2, &cs);
                    case_hist[cs]++;
                    if (cs == 1 || cs == 2) mod_hist[mod2 ? 1 : 0]++;
                    if (!okts) for (m = 0; m < 7; m++) t_scaled[m] = t_ltp[m];
                    if (!final_state(&P, t_scaled, dir, mod2, q_0, v_0, a_0, v_drive, &qe, &ve, &ae)) { fails++; continue; }
                    err = fabs(qe - q_goal);
                    (*n_checks)++;
                    if (!(err <= tol_q)) fails++;
                    if (err > *worst_err) *worst_err = err;
                    *sum_err += err;
                    (*n_err)++;
                    if (fabs(t_ltp[6] + incr[l] - t_scaled[6]) > tol_t) {
                        (*n_checks) += 3;
                        if (!(fabs(ve) <= tol_q)) fails++;
                        if (!(fabs(ae) <= tol_q)) fails++;
                        if (!(fabs(t_scaled[2] - t_scaled[6]) <= eps)) fails++;
                    }
                }
            }
        }
    }
    return fails;
}

long ltpo_kat_grid_time_scaling(long *n_checks, double *worst_err, long *case_hist)
{
    long mod_hist[2], n_err;
    double sum_err;
    return ltpo_kat_grid_time_scaling_stats(n_checks, worst_err, case_hist, mod_hist, &sum_err, &n_err);
}

/*
 * The MATLAB original's two grid tests, restated as procedures (the functions under test are the oracle's MATLAB-semantics
 * twins, P.semantics = 1):
 *   tests/gridTestOneJoint.m      every scenario must end within tol = 0.02 of the goal, else it is "not finished"
 *                                 (|v_end| or |a_end| > tol) or a "failure"; the script errors if either list is non-empty
 *   tests/gridTestTimeScaling.m   per scenario and time increment: success, or "not finished" / "failure" / "time error"
 * MATLAB's colon ranges a:step:b are walked as a + k*step (MATLAB's own colon places the last elements from the far end to
 * limit round-off; the scenario grids agree to ~1e-15).
 * out[0..5] = {success, not_finished, failure, time_error, scenarios, matlab flags OR-ed over the run}; *worst_err / *sum_err
 * over |q_end - q_goal|.
 */
int ltpo_matlab_flags(int clear);
static int colon_count(double lo, double step, double hi) { return hi < lo ? 0 : (int)floor((hi - lo) / step * (1.0 + 4e-16) + 1e-9) + 1; }

void ltpo_kat_matlab_grid_one_joint(long *out, double *worst_err, double *sum_err)
{
    const double eps = 1e-6, tol = 0.02, step = 0.1;
    const double q_min[1] = {0}, q_max[1] = {0}, v_max[1] = {1.0}, a_max[1] = {2.0}, j_max[1] = {15.0};
    const double q_0 = 0.5;
    ltpo_planner P = {1, 0.004, q_min, q_max, v_max, a_max, j_max, 1};
    int i, j, k, nq = colon_count(-6, step, 7), nv = colon_count(-(v_max[0] - eps), step, v_max[0] - eps);
    for (i = 0; i < 6; i++) out[i] = 0;
    *worst_err = 0;
    *sum_err = 0;
    ltpo_matlab_flags(1);
    for (i = 0; i < nq; i++) {
        const double q_goal = -6 + i * step;
        for (j = 0; j < nv; j++) {
            const double v_0 = -(v_max[0] - eps) + j * step;
            double a_lb, a_ub;
            int na;
            if (v_0 >= 0) {
                a_lb = -(a_max[0] - eps);
                a_ub = dmin(a_max[0] - eps, sqrt(2 * j_max[0] * (v_max[0] - v_0)));
            } else {
                a_lb = dmax(-(a_max[0] - eps), -sqrt(2 * j_max[0] * (v_max[0] - fabs(v_0))));
                a_ub = a_max[0];
            }
            na = colon_count(a_lb, step, a_ub);
            for (k = 0; k < na; k++) {
                const double a_0 = a_lb + k * step;
                double t[7] = {0, 0, 0, 0, 0, 0, 0}, dir, qe = 0, ve = 0, ae = 0, err;
                char mod;
                out[4]++;
                if (!ltpo_opt_switch_times(&P, 0, q_goal, q_0, v_0, a_0, v_max[0], t, &dir, &mod)) { out[2]++; continue; }   /* MATLAB error() */
                /* gridTestOneJoint.m:42: getTrajectories(t, dir, false, ...) — the standard jerk profile, v_drive = v_max */
                if (!final_state(&P, t, dir, 0, q_0, v_0, a_0, v_max[0], &qe, &ve, &ae)) { out[2]++; continue; }
                err = fabs(qe - q_goal);
                *sum_err += err;
                if (err > *worst_err) *worst_err = err;
                if (err < tol) out[0]++;
                else if (fabs(ve) > tol || fabs(ae) > tol) out[1]++;
                else out[2]++;
            }
        }
    }
    out[5] = ltpo_matlab_flags(1);
}

void ltpo_kat_matlab_grid_time_scaling(long *out, double *worst_err, double *sum_err, long *case_hist)
{
    const double eps = 1e-6, tol_q = 0.02, tol_t = 0.1, step = 0.1;
    const double q_min[1] = {0}, q_max[1] = {0}, v_max[1] = {1.0}, a_max[1] = {2.0}, j_max[1] = {15.0};
    const double q_0 = 0.5;
    const double incr[6] = {0.05, 0.1, 0.2, 0.5, 1.0, 2.0};
    ltpo_planner P = {1, 0.004, q_min, q_max, v_max, a_max, j_max, 1};
    int i, j, k, l, c, nq = colon_count(-6, step, 7), nv = colon_count(-(v_max[0] - eps), step, v_max[0] - eps);
    for (i = 0; i < 6; i++) out[i] = 0;
    for (c = 0; c < 9; c++) case_hist[c] = 0;
    *worst_err = 0;
    *sum_err = 0;
    ltpo_matlab_flags(1);
    for (i = 0; i < nq; i++) {
        const double q_goal = -6 + i * step;
        for (j = 0; j < nv; j++) {
            const double v_0 = -(v_max[0] - eps) + j * step;
            double a_lb, a_ub;
            int na;
            if (v_0 >= 0) {
                a_lb = -(a_max[0] - eps);
                a_ub = dmin(a_max[0] - eps, sqrt(2 * j_max[0] * (v_max[0] - v_0)));
            } else {
                a_lb = dmax(-(a_max[0] - eps), -sqrt(2 * j_max[0] * (v_max[0] - fabs(v_0))));
                a_ub = a_max[0];
            }
            na = colon_count(a_lb, step, a_ub);
            for (k = 0; k < na; k++) {
                const double a_0 = a_lb + k * step;
                double t[7] = {0, 0, 0, 0, 0, 0, 0}, dir;
                char mod;
                if (!ltpo_opt_switch_times(&P, 0, q_goal, q_0, v_0, a_0, v_max[0], t, &dir, &mod)) { out[2]++; out[4]++; continue; }
                for (l = 0; l < 6; l++) {
                    double ts[7] = {0, 0, 0, 0, 0, 0, 0}, v_drive, qe = 0, ve = 0, ae = 0, err;
                    char mod2 = 0;
                    int cs = 0, m, any = 0;
                    if (t[6] < tol_q) break;
                    out[4]++;
                    ltpo_time_scaling_ex(&P, 0, q_goal, q_0, v_0, a_0, dir, t[6] + incr[l], ts, &v_drive, &mod2, &cs);
                    if (ltpo_matlab_flags(0) & 2) { out[2]++; ltpo_matlab_flags(1); out[5] |= 2; continue; }                 /* MATLAB error() */
                    case_hist[cs]++;
                    for (m = 0; m < 7; m++) if (ts[m] != 0.0) any = 1;
                    if (!any) for (m = 0; m < 7; m++) ts[m] = t[m];                                                          /* :54-56 */
                    if (!final_state(&P, ts, dir, mod2, q_0, v_0, a_0, v_drive, &qe, &ve, &ae)) { out[2]++; continue; }
                    err = fabs(qe - q_goal);
                    *sum_err += err;
                    if (err > *worst_err) *worst_err = err;
                    if (err < tol_q && fabs(t[6] + incr[l] - ts[6]) < tol_t) out[0]++;
                    else if (fabs(ve) > tol_q || fabs(ae) > tol_q) out[1]++;
                    else if (err > tol_q) out[2]++;
                    else if (ts[2] == ts[6]) out[0]++;          /* goal reached after maximal braking (correct behaviour) */
                    else out[3]++;
                }
            }
        }
    }
    out[5] |= ltpo_matlab_flags(1);
}
