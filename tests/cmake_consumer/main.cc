// What user code of the reference looks like (README.md usage): construct, plan, read the trajectory.
#include <cstdio>
#include <vector>

#include "long_term_planner/long_term_planner.h"

int main() {
  try {
    long_term_planner::LongTermPlanner ltp(2, 0.004, {-3.1, -3.1}, {3.1, 3.1}, {1.0, 1.0}, {2.0, 2.0}, {15.0, 15.0});
    long_term_planner::Trajectory traj;
    const bool ok = ltp.planTrajectory({1.0, -0.5}, {0.0, 0.0}, {0.1, 0.0}, {0.0, 0.2}, traj);
    std::printf("planTrajectory: %s, length %d\n", ok ? "true" : "false", traj.length);
    return ok && traj.length > 1 ? 0 : 1;
  } catch (const std::exception& e) {
    std::printf("no device: %s\n", e.what());   // built and linked fine; the planner itself needs an MI355X
    return 3;
  }
}
