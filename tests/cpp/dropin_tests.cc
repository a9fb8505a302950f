// tests/cpp/dropin_tests.cc — exercises the drop-in C++ class the way the reference's own gtests do.
//
// The scenarios, tables and tolerances are those of /root/reference/tests/src/long_term_planner_tests.cc
// (OptBrakingTest :12-50, OptSwitchTimesTest :52-109, TrajectoryTestV0/V1/V2 :111-196, TimeScalingTest
// :198-262, gridTestOneJoint :264-323 on a coarser grid) and reach the protected methods through a
// `using`-exporting subclass exactly as tests/include/long_term_planner_fixture.h:34-39 does, so this file
// also proves that the protected signatures are source-compatible. Runs on the GPU (no CPU path exists).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <thread>
#include <vector>

#include "long_term_planner/long_term_planner.h"
#include "long_term_planner/roots.h"

namespace ltpn = long_term_planner;

class LongTermPlannerExposed : public ltpn::LongTermPlanner {
 public:
  using LongTermPlanner::getTrajectory;
  using LongTermPlanner::optBraking;
  using LongTermPlanner::optSwitchTimes;
  using LongTermPlanner::timeScaling;
  LongTermPlannerExposed() {}
  explicit LongTermPlannerExposed(int dof, double t_sample, std::vector<double> q_min, std::vector<double> q_max,
                                  std::vector<double> v_max, std::vector<double> a_max, std::vector<double> j_max)
      : LongTermPlanner(dof, t_sample, q_min, q_max, v_max, a_max, j_max) {}
};

static int g_checks = 0, g_fails = 0;
#define EXPECT_TRUE(c) do { ++g_checks; if (!(c)) { ++g_fails; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #c); } } while (0)
#define EXPECT_NEAR(a, b, tol) do { ++g_checks; const double a_ = (a), b_ = (b); if (!(std::fabs(a_ - b_) <= (tol))) { ++g_fails; std::printf("FAIL %s:%d  |%s - %s| = |%.12g - %.12g| > %g\n", __FILE__, __LINE__, #a, #b, a_, b_, (double)(tol)); } } while (0)

static LongTermPlannerExposed fixture1dof()
{
  // tests/include/long_term_planner_fixture.h:72-81
  return LongTermPlannerExposed(1, 0.001, {-3.1}, {3.1}, {10}, {2}, {4});
}

static void testOptBraking()
{
  LongTermPlannerExposed ltp = fixture1dof();
  const double eps = 0.01;
  std::vector<double> v_0 = {0, -1.875, -1.875, -0.875, -0.875, 0.5};
  std::vector<double> a_0 = {0, 1, -1, 1, -1, -2};
  std::vector<double> a_max = {2, 2, 2, 4, 4, 4}, j_max = {4, 4, 4, 4, 4, 2};
  std::vector<double> q_goal = {0, -1.0104, -1.9896, -0.2604, -0.7396, -0.4167};
  std::vector<std::vector<double>> t_rel = {{0, 0, 0}, {0.25, 0.5, 0.5}, {0.75, 0.5, 0.5}, {0.25, 0, 0.5}, {0.75, 0, 0.5}, {1.5, 0, 0.5}};
  for (int i = 0; i < 6; i++) {
    ltp.setLimits({-3.1}, {3.1}, {10}, {a_max[i]}, {j_max[i]});
    double q_ltp, dir;
    std::array<double, 7> t_ltp{};
    EXPECT_TRUE(ltp.optBraking(0, v_0[i], a_0[i], q_ltp, t_ltp, dir));
    for (int j = 0; j < 3; j++) EXPECT_NEAR(t_ltp[j], t_rel[i][j], eps);
    EXPECT_NEAR(q_ltp, q_goal[i], eps);
    if (i == 0) continue;
    EXPECT_TRUE(ltp.optBraking(0, -v_0[i], -a_0[i], q_ltp, t_ltp, dir));
    for (int j = 0; j < 3; j++) EXPECT_NEAR(t_ltp[j], t_rel[i][j], eps);
    EXPECT_NEAR(q_ltp, -q_goal[i], eps);
  }
}

static void testOptSwitchTimes()
{
  LongTermPlannerExposed ltp = fixture1dof();
  const double eps = 0.001;
  std::vector<double> v_max = {2, 2, 2, 1, 1, 8, 8, 8, 8};
  std::vector<double> q_goal = {-1.0, 2.927, 2.8854, 0.2396, 0.6354, 1.927, 1.8854, -0.2604, 0.1354};
  std::vector<double> v_0 = {0.0, 0.625, 1.875, -0.875, 0.875, 0.625, 1.875, -0.875, 0.875};
  std::vector<double> a_0 = {0.0, 1.0, -1.0, 1.0, -1.0, 1.0, -1.0, 1.0, -1.0};
  std::vector<std::vector<double>> t = {{0, 0, 0, 0, 0, 0, 0}, {0.25, 0.5, 1.0, 1.5, 2.0, 2.5, 3.0}, {0.5, 0.5, 0.75, 1.25, 1.75, 2.25, 2.75},
      {0.25, 0.75, 1.25, 1.75, 2.25, 2.25, 2.75}, {0.5, 0.5, 0.75, 1.25, 1.75, 1.75, 2.25}, {0.25, 0.5, 1.0, 1.0, 1.5, 2.0, 2.5},
      {0.5, 0.5, 0.5, 0.5, 1.25, 1.75, 2.25}, {0.25, 0.75, 1.25, 1.25, 1.75, 1.75, 2.25}, {0.5, 0.5, 0.5, 0.5, 1.25, 1.25, 1.75}};
  for (int i = 0; i < 9; i++) {
    ltp.setLimits({-3.1}, {3.1}, {v_max[i]}, {2}, {4});
    std::array<double, 7> t_ltp{};
    double dir;
    char mod;
    EXPECT_TRUE(ltp.optSwitchTimes(0, q_goal[i], -1.0, v_0[i], a_0[i], v_max[i], t_ltp, dir, mod));
    for (int j = 0; j < 3; j++) EXPECT_NEAR(t_ltp[j], t[i][j], eps);
    if (i == 0) continue;
    EXPECT_TRUE(ltp.optSwitchTimes(0, -q_goal[i], 1.0, -v_0[i], -a_0[i], v_max[i], t_ltp, dir, mod));
    for (int j = 0; j < 3; j++) EXPECT_NEAR(t_ltp[j], t[i][j], eps);
  }
}

static void testTrajectories()
{
  LongTermPlannerExposed ltp = fixture1dof();
  std::vector<double> v_max = {2, 2, 2, 1, 1, 8, 8, 8, 8};
  std::vector<std::vector<double>> goals = {{1.1, 1.01, 1.05, 1.1, 1.15, 1.2, 1.25, 1.3, 1.5}, {1.0, 1.01, 1.05, 1.1, 1.15, 1.2, 1.25, 1.3, 1.5},
                                            {1.1, 1.1, 1.1, 1.1, 1.1, 1.1, 1.1, 1.1, 1.1}};
  std::vector<std::vector<double>> a0s = {{1e-8, -1e-8, -1e-8, -1e-8, -1e-8, -1e-8, -1e-8, -1e-8, -1e-8},
                                          {1e-8, 1e-8, 1e-8, 1e-8, 1e-8, 1e-8, 1e-8, 1e-8, 1e-8},
                                          {1e-1, 1e-2, 1e-3, 1e-4, 1e-5, 1e-6, 1e-7, 1e-8, 1e-9}};
  for (int v = 0; v < 3; v++)
    for (int i = 0; i < 9; i++) {
      ltp.setLimits({-3.1}, {3.1}, {v_max[i]}, {2}, {4});
      ltpn::Trajectory traj;
      const bool success = ltp.planTrajectory({goals[v][i]}, {1.0}, {0.0}, {a0s[v][i]}, traj);
      EXPECT_TRUE(success);
      if (!success) continue;
      EXPECT_TRUE(traj.dof == 1 && traj.length >= 1 && (int)traj.q[0].size() == traj.length);
      EXPECT_NEAR(traj.q[0][traj.length - 1], goals[v][i], 1e-2);
    }
}

static void testTimeScaling()
{
  LongTermPlannerExposed ltp = fixture1dof();
  const double eps = 0.1;
  std::vector<double> a_max = {2, 2, 2, 2, 2, 2, 2, 2, 2, 4, 4, 4}, j_max = {4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 4, 2};
  std::vector<double> q_goal = {-1.0, 2.927, 2.8854, 0.2396, 0.6354, -7.0104, -8.9896, -3.896, -7.9433, -5.1746, -6.6538, -8.4167};
  std::vector<double> v_0 = {0.0, 0.625, 1.875, -0.875, 0.875, -3.875, -3.875, -1.875, -1.875, -2.875, -2.875, -1.5};
  std::vector<double> a_0 = {0.0, 1, -1, 1, -1, 1, -1, 1, -2, 1, -1, -2};
  std::vector<double> dir = {1.0, 1, 1, 1, 1, -1, -1, -1, -1, -1, -1, -1};
  std::vector<std::vector<double>> t = {{0, 0, 0}, {0.25, 0.5, 1}, {0.5, 0.5, 0.75}, {0.25, 0.75, 1.25}, {0.5, 0.5, 0.75}, {0.25, 0.75, 1.25},
      {0.75, 1.25, 1.75}, {0.25, 0.5, 1}, {0.75, 0.75, 1}, {0.25, 0.25, 0.75}, {0.75, 0.75, 1.25}, {1.5, 1.5, 2}};
  std::vector<double> t_required = {0, 3, 2.75, 2.75, 2.25, 3.25, 3.75, 5, 3.9997, 2.6642, 3.1642, 4.5};
  for (int i = 0; i < 12; i++) {
    ltp.setLimits({-3.1}, {3.1}, {4}, {a_max[i]}, {j_max[i]});
    std::array<double, 7> t_ltp{};
    double v_drive;
    char mod;
    EXPECT_TRUE(ltp.timeScaling(0, q_goal[i], -1.0, v_0[i], a_0[i], dir[i], t_required[i], t_ltp, v_drive, mod));
    for (int j = 0; j < 3; j++) EXPECT_NEAR(t_ltp[j], t[i][j], eps);
    if (i == 0) continue;
    EXPECT_TRUE(ltp.timeScaling(0, -q_goal[i], 1.0, -v_0[i], -a_0[i], -dir[i], t_required[i], t_ltp, v_drive, mod));
    for (int j = 0; j < 3; j++) EXPECT_NEAR(t_ltp[j], t[i][j], eps);
  }
}

static void testGridOneJointCoarse()
{
  // gridTestOneJoint (:264-323) with every 5th grid point (one call = several GPU round trips)
  LongTermPlannerExposed ltp = fixture1dof();
  const double eps = 1e-6, tol = 0.02, step = 0.1, q_0 = 0.5;
  std::vector<double> v_max = {1.0};
  ltp.setSampleTime(0.004);
  ltp.setLimits({-3.1}, {3.1}, v_max, {2.0}, {15.0});
  for (int i = -30; i <= 30; i += 5)
    for (int j = -10; j < 10; j += 5) {
      const double q_goal = i * step, v_0 = j * step;
      for (int k = -10; k < 10; k += 5) {
        const double a_0 = k * step - eps;
        std::array<double, 7> t_ltp{};
        double dir;
        char mod;
        EXPECT_TRUE(ltp.optSwitchTimes(0, q_goal, q_0, v_0, a_0, v_max[0], t_ltp, dir, mod));
        ltpn::Trajectory traj = ltp.getTrajectory({t_ltp}, {dir}, {mod}, {q_0}, {v_0}, {a_0}, v_max);
        EXPECT_TRUE(traj.length >= 1);
        if (traj.length >= 1) EXPECT_NEAR(traj.q[0][traj.length - 1], q_goal, tol);
      }
    }
}

static void testApiSurface()
{
  // default-constructed planner: dof 0 -> planTrajectory false (slowest_joint == -1, cc:39), traj untouched
  ltpn::LongTermPlanner dummy;
  ltpn::Trajectory traj;
  traj.length = -7;
  EXPECT_TRUE(!dummy.planTrajectory({}, {}, {}, {}, traj));
  EXPECT_TRUE(traj.length == -7);
  // checkInputs (cc:68-77), setDoF(double), copy semantics, 6-DoF fixture (fixture.h:97-109)
  LongTermPlannerExposed six(6, 0.001, std::vector<double>(6, -3.1), std::vector<double>(6, 3.1), std::vector<double>(6, 10.0),
                             {2, 2, 2, 4, 4, 4}, {4, 4, 4, 4, 4, 2});
  EXPECT_TRUE(six.checkInputs(std::vector<double>(6, 0.0), std::vector<double>(6, 1.0), std::vector<double>(6, 1.0)));
  EXPECT_TRUE(!six.checkInputs(std::vector<double>(6, 3.2), std::vector<double>(6, 0.0), std::vector<double>(6, 0.0)));
  LongTermPlannerExposed copy = six;
  ltpn::Trajectory t6;
  const bool ok = copy.planTrajectory({1, -1, 0.5, 2, -2, 0}, {0, 0, 0, 0, 0, 0}, {0.1, 0, -0.1, 0, 0.2, 0}, {0, 0.1, 0, -0.1, 0, 0}, t6);
  EXPECT_TRUE(ok);
  EXPECT_TRUE(t6.dof == 6 && (int)t6.q.size() == 6 && (int)t6.j[5].size() == t6.length);
  const double goals[6] = {1, -1, 0.5, 2, -2, 0};
  for (int i = 0; i < 6 && ok; i++) {
    EXPECT_NEAR(t6.q[i][t6.length - 1], goals[i], 0.02);
    EXPECT_NEAR(t6.v[i][t6.length - 1], 0.0, 1e-12);
    EXPECT_NEAR(t6.a[i][t6.length - 1], 0.0, 1e-12);
  }
  // invalid start state: false, trajectory untouched (cc:14-15)
  ltpn::Trajectory keep;
  keep.length = 123;
  EXPECT_TRUE(!six.planTrajectory({1, -1, 0.5, 2, -2, 0}, {9, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}, keep));
  EXPECT_TRUE(keep.length == 123);
  // batched overload
  ltpn::BatchTrajectory b;
  std::vector<double> qg = {1, -1, 0.5, 2, -2, 0, 0.3, 0.2, 0.1, 0, -0.1, -0.2}, z(12, 0.0);
  const long long n_ok = six.planTrajectoryBatch(2, qg.data(), z.data(), z.data(), z.data(), b);
  EXPECT_TRUE(n_ok == 2 && b.length[0] > 1 && b.length[1] > 1);
  EXPECT_NEAR(b.row(1, 0, 0)[b.length[1] - 1], 0.3, 0.02);
  EXPECT_TRUE(b.trajectory(0).length == b.length[0]);
  // NEW: envelopes instead of dense trajectories through the host-pointer class API
  {
    std::vector<double> env;
    ltpn::BatchTrajectory rec;
    const int W = 64, K = 60;
    const long long ok_env = six.planEnvelopeBatch(2, qg.data(), z.data(), z.data(), z.data(), W, K, env, &rec);
    EXPECT_TRUE(ok_env == 2 && rec.length[0] == b.length[0] && (long long)env.size() == 2LL * 6 * K * 2);
    for (int p = 0; p < 2; ++p)
      for (int j = 0; j < 6; ++j)
        for (int w = 0; w < K; ++w) {
          const int L = b.length[p], lo_i = std::min(w * W, L - 1), hi_i = std::min(w * W + W, L);
          double mn = b.row(p, 0, j)[lo_i], mx = mn;
          for (int i = lo_i; i < std::max(hi_i, lo_i + 1); ++i) { mn = std::min(mn, b.row(p, 0, j)[i]); mx = std::max(mx, b.row(p, 0, j)[i]); }
          EXPECT_TRUE(env[((p * 6 + j) * K + w) * 2] == mn && env[((p * 6 + j) * K + w) * 2 + 1] == mx);
        }
  }
  // NEW options travel with copies and survive a handle that is created later
  LongTermPlannerExposed opts(6, 0.001, std::vector<double>(6, -3.1), std::vector<double>(6, 3.1), std::vector<double>(6, 10.0),
                              {2, 2, 2, 4, 4, 4}, {4, 4, 4, 4, 4, 2});
  opts.setMaxSamples(100);      // before any device call: no handle exists yet
  opts.setGoalCheck(true);
  LongTermPlannerExposed opts2 = opts;
  ltpn::BatchTrajectory bc;
  std::vector<double> far = qg;
  far[6] = 5.0;                 // query 1: goal outside [-3.1, 3.1]
  opts2.planTrajectoryBatch(2, far.data(), z.data(), z.data(), z.data(), bc);
  EXPECT_TRUE(bc.stored[0] == 100 && bc.length[0] == b.length[0]);
  EXPECT_TRUE((bc.status[1] & LTP_STATUS_GOAL_OUTSIDE) != 0 && bc.length[1] == 0);
  for (int k = 0; k < 100; k += 9) EXPECT_TRUE(bc.row(0, 0, 2)[k] == b.row(0, 0, 2)[k]);
  opts2.setMaxSamples(0);
  opts2.setGoalCheck(false);
  opts2.planTrajectoryBatch(2, far.data(), z.data(), z.data(), z.data(), bc);
  EXPECT_TRUE(bc.stored[0] == b.length[0] && bc.status[1] == LTP_STATUS_END_LIMIT);
}

// The reference's root-finder test (tests/src/roots_tests.cc:9-32) on the device path: float, degree 6, the six eigenvalues
// in Eigen's output order at the reference's tolerances.
static void testRootsHeader()
{
  const std::vector<float> poly = {144.f, -1008.f, 2448.f, 3024.01920000000f, -15768.1344000000f, 0.f, 22752.4032012800f};
  const std::vector<std::complex<float>> r = ltpn::roots<float>(poly);
  EXPECT_TRUE(r.size() == 6);
  const double re[6] = {-1.67276, -1.35687, 2.00001, 2.09261, 2.9685, 2.9685}, im[6] = {0, 0, 0, 0, 2.79663, -2.79663};
  for (int i = 0; i < 6 && r.size() == 6; ++i) {
    EXPECT_NEAR(r[i].real(), re[i], 1e-5);
    EXPECT_NEAR(r[i].imag(), im[i], i < 4 ? 1e-9 : 1e-5);
  }
  EXPECT_NEAR(ltpn::getSmallestPositiveNonComplexRoot<float>(r), 2.00001, 1e-5);
  // double: (x - 1)(x - 2)(x^2 + 1) -> smallest admissible root 1; x^4 + 1 -> none
  EXPECT_NEAR(ltpn::getSmallestPositiveNonComplexRoot<double>(ltpn::roots<double>({1, -3, 3, -3, 2})), 1.0, 1e-12);
  EXPECT_TRUE(std::isinf(ltpn::getSmallestPositiveNonComplexRoot<double>(ltpn::roots<double>({1, 0, 0, 0, 1}))));
}

// Two threads issue the FIRST call on one object at the same time (the reference allows concurrent planTrajectory on
// one object, SURVEY §8(b) "Threading"): the lazily created device handle must be created once, and both get results.
static void testConcurrentFirstCall()
{
  for (int round = 0; round < 4; ++round) {
    LongTermPlannerExposed ltp(6, 0.001, std::vector<double>(6, -3.1), std::vector<double>(6, 3.1), std::vector<double>(6, 10.0),
                               {2, 2, 2, 4, 4, 4}, {4, 4, 4, 4, 4, 2});
    ltpn::Trajectory t[2];
    bool ok[2] = {false, false};
    std::string err[2];
    auto work = [&](int w) {
      try {
        ok[w] = ltp.planTrajectory({1, -1, 0.5, 2, -2, 0}, {0, 0, 0, 0, 0, 0}, {0.1, 0, -0.1, 0, 0.2, 0}, {0, 0.1, 0, -0.1, 0, 0}, t[w]);
      } catch (const std::exception& e) { err[w] = e.what(); }
    };
    std::thread a(work, 0), b(work, 1);
    a.join(); b.join();
    EXPECT_TRUE(err[0].empty() && err[1].empty());
    EXPECT_TRUE(ok[0] && ok[1] && t[0].length == t[1].length && t[0].length > 1);
    if (ok[0] && ok[1] && t[0].length == t[1].length)
      for (int i = 0; i < 6; ++i) EXPECT_TRUE(std::memcmp(t[0].q[i].data(), t[1].q[i].data(), sizeof(double) * t[0].length) == 0);
  }
}

// NEW: one process, one handle per shard; here three virtual shards on device 0 (and an odd batch size, so the
// shards differ in length). Everything must have the bits of the unsharded call.
static void testShardedBatch()
{
  LongTermPlannerExposed ltp(6, 0.004, std::vector<double>(6, -3.1), std::vector<double>(6, 3.1), std::vector<double>(6, 1.0),
                             {2, 2, 2, 4, 4, 4}, {15, 15, 15, 15, 15, 8});
  const long long n = 101;
  std::vector<double> qg(n * 6), q0(n * 6), v0(n * 6), a0(n * 6);
  unsigned long long z = 88172645463325252ull;
  auto u = [&]() { z ^= z << 13; z ^= z >> 7; z ^= z << 17; return (double)(z >> 11) * (1.0 / 9007199254740992.0); };
  for (long long i = 0; i < n * 6; ++i) { qg[i] = -3.0 + 6.0 * u(); q0[i] = -3.0 + 6.0 * u(); v0[i] = -0.5 + u(); a0[i] = -0.5 + u(); }
  q0[5 * 6 + 2] = 7.0;          // one invalid query (checkInputs false) inside shard 0
  ltpn::BatchTrajectory one, many, none;
  const long long ok1 = ltp.planTrajectoryBatch(n, qg.data(), q0.data(), v0.data(), a0.data(), one);
  const long long ok3 = ltp.planTrajectoryBatchSharded(n, qg.data(), q0.data(), v0.data(), a0.data(), many, {0, 0, 0});
  EXPECT_TRUE(ok1 == ok3 && ok1 > 0 && ok1 < n);
  EXPECT_TRUE(one.status == many.status && one.length == many.length && one.stored == many.stored && one.slowest == many.slowest);
  EXPECT_TRUE(one.offsets == many.offsets && one.mod == many.mod);
  auto same = [](const std::vector<double>& a, const std::vector<double>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), sizeof(double) * a.size()) == 0);
  };
  EXPECT_TRUE(same(one.t_opt, many.t_opt) && same(one.t_scaled, many.t_scaled) && same(one.dir, many.dir));
  EXPECT_TRUE(same(one.v_drive, many.v_drive) && same(one.t_required, many.t_required));
  EXPECT_TRUE(same(one.packed, many.packed) && !one.packed.empty());
  // switching times only: status still is planTrajectory's verdict (end-limit check without sampling, cc:59-61)
  const long long ok0 = ltp.planTrajectoryBatchSharded(n, qg.data(), q0.data(), v0.data(), a0.data(), none, {0, 0}, false);
  EXPECT_TRUE(ok0 == ok1 && none.status == one.status && none.packed.empty() && none.offsets == one.offsets);
  // more shards than queries: the tail shards are empty
  ltpn::BatchTrajectory tiny, tiny1;
  ltp.planTrajectoryBatchSharded(2, qg.data(), q0.data(), v0.data(), a0.data(), tiny, {0, 0, 0, 0});
  ltp.planTrajectoryBatch(2, qg.data(), q0.data(), v0.data(), a0.data(), tiny1);
  EXPECT_TRUE(tiny.status == tiny1.status && same(tiny.packed, tiny1.packed) && tiny.offsets == tiny1.offsets);
  // the envelope consumer over shards (ltp_plan_envelope_multi_host): envelopes and records of the unsharded call, bit for bit
  std::vector<double> env1, env3, env9;
  ltpn::BatchTrajectory r1, r3;
  const long long e1 = ltp.planEnvelopeBatch(n, qg.data(), q0.data(), v0.data(), a0.data(), 25, 9, env1, &r1);
  const long long e3 = ltp.planEnvelopeBatchSharded(n, qg.data(), q0.data(), v0.data(), a0.data(), 25, 9, env3, {0, 0, 0}, &r3);
  EXPECT_TRUE(e1 == e3 && e1 == ok1 && env1.size() == (size_t)n * 6 * 9 * 2);
  EXPECT_TRUE(env1.size() == env3.size() && std::memcmp(env1.data(), env3.data(), sizeof(double) * env1.size()) == 0);   // NaN rows of the failed plan included
  EXPECT_TRUE(r1.status == r3.status && r1.length == r3.length && same(r1.t_scaled, r3.t_scaled) && r1.status == one.status);
  ltp.planEnvelopeBatchSharded(3, qg.data(), q0.data(), v0.data(), a0.data(), 25, 9, env9, {0, 0, 0, 0, 0});   // empty tail shards
  EXPECT_TRUE(env9.size() == 3u * 6 * 9 * 2 && std::memcmp(env9.data(), env1.data(), sizeof(double) * env9.size()) == 0);
}

// Round 5 options of the drop-in class: the pow rule (default: glibc's pow restated; setLibmPow(false): correctly rounded powers) and
// the analytic envelopes. Copies inherit both; the rules agree to well within 1e-9 and differ in some last bits; the analytic envelopes
// equal the exhaustive ones (their candidates are samples of the row).
static void testRound5Options()
{
  const int n = 400;
  std::vector<double> qg(n * 6), q0(n * 6), v0(n * 6, 0.0), a0(n * 6, 0.0);
  unsigned long long st = 88172645463325252ull;
  auto u = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) * (1.0 / 9007199254740992.0); };
  for (int i = 0; i < n * 6; ++i) { qg[i] = -3.0 + 6.0 * u(); q0[i] = -3.0 + 6.0 * u(); v0[i] = -0.9 + 1.8 * u(); }
  ltpn::LongTermPlanner libm(6, 0.004, std::vector<double>(6, -3.14), std::vector<double>(6, 3.14), std::vector<double>(6, 1.0), std::vector<double>(6, 2.0),
                            std::vector<double>(6, 15.0));
  ltpn::LongTermPlanner exact = libm;                       // copies carry the options
  exact.setLibmPow(false);
  ltpn::LongTermPlanner exact2 = exact;
  ltpn::BatchTrajectory a, b, c;
  const long long oa = libm.planTrajectoryBatch(n, qg.data(), q0.data(), v0.data(), a0.data(), a, false);
  const long long ob = exact.planTrajectoryBatch(n, qg.data(), q0.data(), v0.data(), a0.data(), b, false);
  const long long oc = exact2.planTrajectoryBatch(n, qg.data(), q0.data(), v0.data(), a0.data(), c, false);
  EXPECT_TRUE(oa == ob && ob == oc && a.status == b.status && a.length == b.length);
  EXPECT_TRUE(b.t_scaled.size() == c.t_scaled.size() && std::memcmp(b.t_scaled.data(), c.t_scaled.data(), sizeof(double) * b.t_scaled.size()) == 0);
  double worst = 0.0;
  size_t other_bits = 0;
  for (size_t i = 0; i < a.t_scaled.size(); ++i) {
    const double d = std::fabs(a.t_scaled[i] - b.t_scaled[i]);
    if (d == d && d > worst) worst = d;
    other_bits += std::memcmp(&a.t_scaled[i], &b.t_scaled[i], sizeof(double)) != 0;
  }
  std::printf("pow rules, %d plans: switching times differ by at most %.3e s, %zu of %zu entries in their bits\n", n, worst, other_bits, a.t_scaled.size());
  EXPECT_TRUE(worst < 1e-9);
  std::vector<double> e_all, e_ana;
  libm.setAnalyticEnvelopes(false);                         // the exhaustive form (the default is the analytic one since round 6)
  const long long e1 = libm.planEnvelopeBatch(n, qg.data(), q0.data(), v0.data(), a0.data(), 20, 12, e_all);
  libm.setAnalyticEnvelopes(true);
  const long long e2 = libm.planEnvelopeBatch(n, qg.data(), q0.data(), v0.data(), a0.data(), 20, 12, e_ana);
  EXPECT_TRUE(e1 == e2 && e_all.size() == e_ana.size() && e_all.size() == (size_t)n * 6 * 12 * 2);
  double wenv = 0.0;
  for (size_t i = 0; i < e_all.size(); ++i) {
    if (e_all[i] != e_all[i]) { EXPECT_TRUE(e_ana[i] != e_ana[i]); continue; }
    wenv = std::fmax(wenv, std::fabs(e_all[i] - e_ana[i]));
  }
  EXPECT_TRUE(wenv <= 1e-12);
}

int main()
{
  try {
    testOptBraking();
    testOptSwitchTimes();
    testTrajectories();
    testTimeScaling();
    testGridOneJointCoarse();
    testApiSurface();
    testRootsHeader();
    testConcurrentFirstCall();
    testShardedBatch();
    testRound5Options();
  } catch (const std::exception& e) {
    std::printf("EXCEPTION: %s\n", e.what());
    return 2;
  }
  std::printf("%d checks, %d failures\n", g_checks, g_fails);
  return g_fails ? 1 : 0;
}
