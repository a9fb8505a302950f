// long_term_planner/long_term_planner.h — drop-in for the reference header of the same path.
//
// Same namespace, struct, class, constructor and method signatures as
// /root/reference/include/long_term_planner/long_term_planner.h (Trajectory :37-45, sign :54-56,
// LongTermPlanner :61-308; implementation src/long_term_planner.cc). Every method forwards to the
// MI355X library through the C ABI of include/ltp_hip.h — there is no host implementation of the
// planner arithmetic in this header or behind it, and no <Eigen/Dense> dependency (the reference
// header pulls Eigen in publicly; the root finder now lives on the device).
//
// Differences a user can observe:
//   * a HIP device is required; a missing device / HIP error throws std::runtime_error (the reference
//     has no error channel besides `bool`, and silently computing on the CPU is not offered);
//   * the reference's std::cerr diagnostic at cc:343 is not printed;
//   * corners the reference leaves undefined (out-of-range writes of the sampler,
//     non-finite switching times) are defined: dropped writes / `false`;
//   * NEW: planTrajectoryBatch(), the batched overload this library exists for; planTrajectoryBatchSharded() and
//     planEnvelopeBatchSharded(), the same over several devices from one process (contiguous query ranges, no collective).
// Threading: as with the reference, concurrent planTrajectory / planTrajectoryBatch calls on one object are allowed as
// long as no setter runs at the same time (the lazily created device handle is guarded by a mutex).
#ifndef long_term_planner_H
#define long_term_planner_H

#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <functional>
#include <math.h>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <string>
#include <vector>

#include "ltp_hip.h"

namespace long_term_planner {

/** @brief Trajectory structure (reference long_term_planner.h:37-45). q/v/a/j are [joint][sample]. */
struct Trajectory {
  int dof;
  double t_sample;
  int length;
  std::vector<std::vector<double>> q;
  std::vector<std::vector<double>> v;
  std::vector<std::vector<double>> a;
  std::vector<std::vector<double>> j;
};

/** @brief -1 / 0 / +1 (reference long_term_planner.h:54-56). */
template <typename T> int sign(T val) {
  return (T(0) < val) - (val < T(0));
}

/** @brief Result of the batched overload: switching-time records plus packed dense trajectories. */
struct BatchTrajectory {
  long long n = 0;
  int dof = 0;
  double t_sample = 0.0;
  std::vector<double> t_opt;        ///< [n][dof][7]
  std::vector<double> t_scaled;     ///< [n][dof][7]
  std::vector<double> dir;          ///< [n][dof]
  std::vector<double> v_drive;      ///< [n][dof]
  std::vector<signed char> mod;     ///< [n][dof]
  std::vector<double> t_required;   ///< [n]
  std::vector<int> slowest;         ///< [n]
  std::vector<int> length;          ///< [n] Trajectory::length, 0 if the plan failed before sampling
  std::vector<int> stored;          ///< [n] samples stored per row: length, or less when setMaxSamples() is in effect
  std::vector<int> status;          ///< [n] LTP_STATUS_* bits; planTrajectory's bool is LongTermPlanner::planOk(status)
  std::vector<unsigned long long> offsets;  ///< [n+1] plan p occupies packed[offsets[p], offsets[p+1])
  std::vector<double> packed;       ///< per plan: [q,v,a,j][joint][ltp_row_stride(length)]
  /// pointer to sample 0 of array `arr` (0=q,1=v,2=a,3=j) of joint `joint` of plan `p`
  const double* row(long long p, int arr, int joint) const {
    return packed.data() + offsets[p] + (static_cast<std::size_t>(arr) * dof + joint) * ltp_row_stride(stored[p]);
  }
  /// copy plan p out as a reference-style Trajectory
  Trajectory trajectory(long long p) const {
    Trajectory t;
    t.dof = dof; t.t_sample = t_sample; t.length = stored[p];
    std::vector<std::vector<double>>* dst[4] = {&t.q, &t.v, &t.a, &t.j};
    for (int arr = 0; arr < 4; ++arr) {
      dst[arr]->resize(dof);
      for (int i = 0; i < dof; ++i) (*dst[arr])[i].assign(row(p, arr, i), row(p, arr, i) + stored[p]);
    }
    return t;
  }
};

/** @brief Plans a trajectory for multiple joints (reference long_term_planner.h:61-308). */
class LongTermPlanner {
 private:
  int dof_;
  double t_sample_;
  std::vector<double> q_min_;
  std::vector<double> q_max_;
  std::vector<double> v_max_;
  std::vector<double> a_max_;
  std::vector<double> j_max_;

  // device-side twins of the members above; created lazily, never shared between copies. handle_ serves the
  // single-device calls, shards_[i] the i-th shard of planTrajectoryBatchSharded. mu_ guards creation/configuration:
  // the reference allows concurrent planTrajectory calls on one object (it mutates no members), so two threads'
  // first calls must not race here either.
  struct DeviceTwin {
    ltp_planner* h = nullptr;
    int device = 0;
    bool dirty = true;
  };
  mutable DeviceTwin handle_;
  mutable std::vector<DeviceTwin> shards_;
  mutable std::mutex mu_;
  int device_ = 0;
  // NEW options (defaults = the reference's behaviour); kept here so that copies and re-created handles inherit them
  int max_samples_ = 0;
  int sample_stride_ = 1;
  bool goal_check_ = false;
  int semantics_ = LTP_SEMANTICS_CPP;
  int pow_rule_ = LTP_POW_LIBM;
  int envelope_mode_ = LTP_ENVELOPE_ANALYTIC;

  static void raise(const ltp_planner* h, int rc, const char* what) {
    throw std::runtime_error(std::string("long_term_planner (MI355X): ") + what + " failed with code " + std::to_string(rc) +
                             (h ? std::string(": ") + ltp_last_error(h) : std::string(" (no HIP device? there is no CPU fallback)")));
  }

  void markDirty() {
    handle_.dirty = true;
    for (auto& s : shards_) s.dirty = true;
  }

  // caller holds mu_
  ltp_planner* ready(DeviceTwin& t, int device) const {
    if (t.h && t.device != device) { ltp_destroy(t.h); t.h = nullptr; }
    if (!t.h) {
      const int rc = ltp_create(0, t_sample_, nullptr, nullptr, nullptr, nullptr, nullptr, device, &t.h);
      if (rc != LTP_OK) raise(nullptr, rc, "ltp_create");
      t.device = device;
      t.dirty = true;
    }
    if (t.dirty) {
      const int n = static_cast<int>(std::min({q_min_.size(), q_max_.size(), v_max_.size(), a_max_.size(), j_max_.size()}));
      int rc = ltp_set_limits(t.h, n, q_min_.data(), q_max_.data(), v_max_.data(), a_max_.data(), j_max_.data());
      if (rc != LTP_OK) raise(t.h, rc, "ltp_set_limits");
      if ((rc = ltp_set_sample_time(t.h, t_sample_)) != LTP_OK) raise(t.h, rc, "ltp_set_sample_time");
      if ((rc = ltp_set_dof(t.h, dof_)) != LTP_OK) raise(t.h, rc, "ltp_set_dof");
      if ((rc = ltp_set_max_samples(t.h, max_samples_)) != LTP_OK) raise(t.h, rc, "ltp_set_max_samples");
      if ((rc = ltp_set_sample_stride(t.h, sample_stride_)) != LTP_OK) raise(t.h, rc, "ltp_set_sample_stride");
      if ((rc = ltp_set_goal_check(t.h, goal_check_ ? 1 : 0)) != LTP_OK) raise(t.h, rc, "ltp_set_goal_check");
      if ((rc = ltp_set_semantics(t.h, semantics_)) != LTP_OK) raise(t.h, rc, "ltp_set_semantics");
      if ((rc = ltp_set_pow_rule(t.h, pow_rule_)) != LTP_OK) raise(t.h, rc, "ltp_set_pow_rule");
      if ((rc = ltp_set_envelope_mode(t.h, envelope_mode_)) != LTP_OK) raise(t.h, rc, "ltp_set_envelope_mode");
      t.dirty = false;
    }
    return t.h;
  }

  ltp_planner* handle() const {
    std::lock_guard<std::mutex> g(mu_);
    return ready(handle_, device_);
  }

  void release() {
    if (handle_.h) ltp_destroy(handle_.h);
    handle_.h = nullptr;
    for (auto& s : shards_)
      if (s.h) ltp_destroy(s.h);
    shards_.clear();
  }

  // sizes `out` for n queries and returns the record pointers the C ABI fills
  ltp_records prepare(long long n, BatchTrajectory& out, double& dummy_d, signed char& dummy_c) const {
    const std::size_t nd = static_cast<std::size_t>(n) * dof_;
    out.n = n; out.dof = dof_; out.t_sample = t_sample_;
    out.t_opt.assign(nd * 7, 0.0); out.t_scaled.assign(nd * 7, 0.0); out.dir.assign(nd, 0.0); out.v_drive.assign(nd, 0.0);
    out.mod.assign(nd, 0); out.t_required.assign(n, 0.0); out.slowest.assign(n, -1); out.length.assign(n, 0);
    out.status.assign(n, 0); out.offsets.assign(n + 1, 0ull); out.packed.clear(); out.stored.assign(n, 0);
    // zero-sized vectors have a null data(); the C ABI wants non-null record pointers
    return ltp_records{nd ? out.t_opt.data() : &dummy_d, nd ? out.t_scaled.data() : &dummy_d, nd ? out.dir.data() : &dummy_d,
                       nd ? out.v_drive.data() : &dummy_d, nd ? out.mod.data() : &dummy_c, out.t_required.data(),
                       out.slowest.data(), out.length.data(), out.status.data()};
  }

  long long finish(ltp_planner* h, long long n, double* packed, BatchTrajectory& out) const {
    if (packed) {
      out.packed.assign(packed, packed + out.offsets[n]);
      ltp_free_host(packed);
    }
    long long ok = 0;
    for (long long p = 0; p < n; ++p) {
      ok += planOk(out.status[p]);
      out.stored[p] = ltp_stored_samples(h, out.length[p]);
    }
    return ok;
  }

 public:
  /** @brief NEW: planTrajectory's bool for a batch status word: every bit but the informational LTP_STATUS_MATLAB_COMPLEX
   *  (the plan is delivered) must be clear. planTrajectory, the batch counters and bench.py all use this rule. */
  static inline bool planOk(int status) { return (status & ~LTP_STATUS_MATLAB_COMPLEX) == 0; }

  /** @brief Dummy planner (reference long_term_planner.h:103-105). */
  LongTermPlanner() : dof_(0), t_sample_(0.001) {}

  /** @brief reference long_term_planner.h:118-131 */
  LongTermPlanner(int dof, double t_sample, std::vector<double> q_min, std::vector<double> q_max, std::vector<double> v_max,
                  std::vector<double> a_max, std::vector<double> j_max)
      : dof_(dof), t_sample_(t_sample), q_min_(q_min), q_max_(q_max), v_max_(v_max), a_max_(a_max), j_max_(j_max) {}

  LongTermPlanner(const LongTermPlanner& o)
      : dof_(o.dof_), t_sample_(o.t_sample_), q_min_(o.q_min_), q_max_(o.q_max_), v_max_(o.v_max_), a_max_(o.a_max_),
        j_max_(o.j_max_), device_(o.device_), max_samples_(o.max_samples_), sample_stride_(o.sample_stride_),
        goal_check_(o.goal_check_), semantics_(o.semantics_), pow_rule_(o.pow_rule_), envelope_mode_(o.envelope_mode_) {}
  LongTermPlanner& operator=(const LongTermPlanner& o) {
    if (this != &o) {
      dof_ = o.dof_; t_sample_ = o.t_sample_; q_min_ = o.q_min_; q_max_ = o.q_max_; v_max_ = o.v_max_; a_max_ = o.a_max_;
      j_max_ = o.j_max_; device_ = o.device_; max_samples_ = o.max_samples_; sample_stride_ = o.sample_stride_;
      goal_check_ = o.goal_check_; semantics_ = o.semantics_; pow_rule_ = o.pow_rule_; envelope_mode_ = o.envelope_mode_; markDirty();
    }
    return *this;
  }
  ~LongTermPlanner() { release(); }

  /** @brief reference long_term_planner.h:144-150, src/long_term_planner.cc:7-63 */
  bool planTrajectory(const std::vector<double>& q_goal, const std::vector<double>& q_0, const std::vector<double>& v_0,
                      const std::vector<double>& a_0, Trajectory& traj) {
    BatchTrajectory b;
    planTrajectoryBatch(1, q_goal.data(), q_0.data(), v_0.data(), a_0.data(), b);
    const int st = b.status[0];
    // the reference leaves `traj` untouched when it returns false before sampling (cc:14-39)
    if (st & (LTP_STATUS_INVALID_INPUT | LTP_STATUS_OPT_FAILED | LTP_STATUS_NO_SLOWEST | LTP_STATUS_NONFINITE | LTP_STATUS_GOAL_OUTSIDE |
              LTP_STATUS_MATLAB_ERROR))
      return false;
    traj = b.trajectory(0);
    return (st & ~LTP_STATUS_MATLAB_COMPLEX) == 0;   // LTP_STATUS_END_LIMIT: false with the trajectory filled (cc:59-61)
  }

  /**
   * @brief NEW batched overload: n independent queries, row-major [n][dof] host arrays.
   * @return number of queries for which planTrajectory would have returned true.
   */
  long long planTrajectoryBatch(long long n, const double* q_goal, const double* q_0, const double* v_0, const double* a_0,
                                BatchTrajectory& out, bool sample = true) {
    ltp_planner* h = handle();
    double dummy_d = 0; signed char dummy_c = 0;
    const ltp_records rec = prepare(n, out, dummy_d, dummy_c);
    double* packed = nullptr;
    const int rc = ltp_plan_batch_host(h, n, q_goal, q_0, v_0, a_0, &rec, out.offsets.data(), sample ? &packed : nullptr);
    if (rc != LTP_OK) raise(h, rc, "ltp_plan_batch_host");
    return finish(h, n, packed, out);
  }

  /**
   * @brief NEW: planTrajectoryBatch over several devices from ONE process. Shard g — the contiguous
   * query range ltp_shard_range(n, g, devices.size()) — is planned on HIP device devices[g] by its own handle and host
   * thread; limits are replicated, nothing is exchanged between devices (queries are independent). `out` is
   * bit-identical to planTrajectoryBatch over all n queries. A device may be listed more than once (virtual shards).
   * @return number of queries for which planTrajectory would have returned true.
   */
  long long planTrajectoryBatchSharded(long long n, const double* q_goal, const double* q_0, const double* v_0, const double* a_0,
                                       BatchTrajectory& out, const std::vector<int>& devices, bool sample = true) {
    if (devices.empty()) throw std::runtime_error("long_term_planner (MI355X): planTrajectoryBatchSharded needs at least one device");
    std::vector<ltp_planner*> hs(devices.size());
    {
      std::lock_guard<std::mutex> g(mu_);
      if (shards_.size() < devices.size()) shards_.resize(devices.size());
      for (std::size_t i = 0; i < devices.size(); ++i) hs[i] = ready(shards_[i], devices[i]);
    }
    double dummy_d = 0; signed char dummy_c = 0;
    const ltp_records rec = prepare(n, out, dummy_d, dummy_c);
    double* packed = nullptr;
    const int rc = ltp_plan_batch_multi(hs.data(), static_cast<int>(hs.size()), n, q_goal, q_0, v_0, a_0, &rec, out.offsets.data(),
                                        sample ? &packed : nullptr);
    if (rc != LTP_OK) raise(hs[0], rc, "ltp_plan_batch_multi");
    return finish(hs[0], n, packed, out);
  }

  /**
   * @brief NEW: plan n queries and reduce every trajectory on the device to per-joint position envelopes —
   * env[((p*dof + joint)*n_windows + w)*2 + {0,1}] = {min, max} of the q samples w*window .. (w+1)*window-1; windows past
   * the end hold the last position, plans without a trajectory NaN. The dense trajectories never exist, so this also
   * works for batches whose trajectories would not fit in memory. `out` (optional) receives the records.
   * @return number of queries for which planTrajectory would have returned true.
   */
  long long planEnvelopeBatch(long long n, const double* q_goal, const double* q_0, const double* v_0, const double* a_0,
                              int window, int n_windows, std::vector<double>& env, BatchTrajectory* out = nullptr) {
    ltp_planner* h = handle();
    BatchTrajectory local;
    BatchTrajectory& b = out ? *out : local;
    const std::size_t nd = static_cast<std::size_t>(n) * dof_;
    b.n = n; b.dof = dof_; b.t_sample = t_sample_;
    b.t_opt.assign(nd * 7, 0.0); b.t_scaled.assign(nd * 7, 0.0); b.dir.assign(nd, 0.0); b.v_drive.assign(nd, 0.0);
    b.mod.assign(nd, 0); b.t_required.assign(n, 0.0); b.slowest.assign(n, -1); b.length.assign(n, 0);
    b.status.assign(n, 0); b.offsets.assign(n + 1, 0ull); b.packed.clear(); b.stored.assign(n, 0);
    env.assign(nd * static_cast<std::size_t>(n_windows > 0 ? n_windows : 0) * 2, 0.0);
    double dummy_d = 0; signed char dummy_c = 0;
    ltp_records rec{nd ? b.t_opt.data() : &dummy_d, nd ? b.t_scaled.data() : &dummy_d, nd ? b.dir.data() : &dummy_d,
                    nd ? b.v_drive.data() : &dummy_d, nd ? b.mod.data() : &dummy_c, b.t_required.data(), b.slowest.data(),
                    b.length.data(), b.status.data()};
    const int rc = ltp_plan_envelope_host(h, n, q_goal, q_0, v_0, a_0, window, n_windows, &rec, env.empty() ? &dummy_d : env.data());
    if (rc != LTP_OK) raise(h, rc, "ltp_plan_envelope_host");
    long long ok = 0;
    for (long long p = 0; p < n; ++p) ok += planOk(b.status[p]);
    return ok;
  }

  /**
   * @brief NEW: planEnvelopeBatch over several devices from ONE process — shard g, the contiguous query
   * range ltp_shard_range(n, g, devices.size()), is planned and reduced on HIP device devices[g] by its own handle and host
   * thread (ltp_plan_envelope_multi_host); `env` and `out` are bit-identical to planEnvelopeBatch over all n queries. This
   * is the sharded call that makes sense at scale: envelopes are 16 bytes per window, the dense rows never leave a device.
   * @return number of queries for which planTrajectory would have returned true.
   */
  long long planEnvelopeBatchSharded(long long n, const double* q_goal, const double* q_0, const double* v_0, const double* a_0,
                                     int window, int n_windows, std::vector<double>& env, const std::vector<int>& devices,
                                     BatchTrajectory* out = nullptr) {
    if (devices.empty()) throw std::runtime_error("long_term_planner (MI355X): planEnvelopeBatchSharded needs at least one device");
    std::vector<ltp_planner*> hs(devices.size());
    {
      std::lock_guard<std::mutex> g(mu_);
      if (shards_.size() < devices.size()) shards_.resize(devices.size());
      for (std::size_t i = 0; i < devices.size(); ++i) hs[i] = ready(shards_[i], devices[i]);
    }
    BatchTrajectory local;
    BatchTrajectory& b = out ? *out : local;
    double dummy_d = 0; signed char dummy_c = 0;
    const ltp_records rec = prepare(n, b, dummy_d, dummy_c);
    const std::size_t nd = static_cast<std::size_t>(n) * dof_;
    env.assign(nd * static_cast<std::size_t>(n_windows > 0 ? n_windows : 0) * 2, 0.0);
    const int rc = ltp_plan_envelope_multi_host(hs.data(), static_cast<int>(hs.size()), n, q_goal, q_0, v_0, a_0, window, n_windows, &rec,
                                                env.empty() ? &dummy_d : env.data());
    if (rc != LTP_OK) raise(hs[0], rc, "ltp_plan_envelope_multi_host");
    long long ok = 0;
    for (long long p = 0; p < n; ++p) ok += planOk(b.status[p]);
    return ok;
  }

  /** @brief reference long_term_planner.h:161-165, cc:68-77 */
  bool checkInputs(const std::vector<double>& q_0, const std::vector<double>& v_0, const std::vector<double>& a_0) {
    int ok = 0;
    ltp_planner* h = handle();
    const int rc = ltp_check_inputs_host(h, q_0.data(), v_0.data(), a_0.data(), &ok);
    if (rc != LTP_OK) raise(h, rc, "ltp_check_inputs_host");
    return ok != 0;
  }

  /** @brief reference long_term_planner.h:176-187 */
  inline void setLimits(std::vector<double> q_min, std::vector<double> q_max, std::vector<double> v_max,
                        std::vector<double> a_max, std::vector<double> j_max) {
    q_min_ = q_min; q_max_ = q_max; v_max_ = v_max; a_max_ = a_max; j_max_ = j_max;
    markDirty();
  }

  /** @brief reference long_term_planner.h:194-196 */
  inline void setSampleTime(double t_sample) { t_sample_ = t_sample; markDirty(); }

  /** @brief reference long_term_planner.h:203-205 (takes a double there as well) */
  inline void setDoF(double dof) { dof_ = dof; markDirty(); }

  /** @brief NEW: store only the first `max_samples` samples of each trajectory (0 = all, the reference's behaviour). */
  inline void setMaxSamples(int max_samples) {
    if (max_samples < 0) throw std::runtime_error("long_term_planner (MI355X): max_samples < 0");
    max_samples_ = max_samples; markDirty();
  }

  /** @brief NEW: store every `stride`-th sample of each trajectory (1 = every sample, the reference's behaviour). */
  inline void setSampleStride(int stride) {
    if (stride < 1) throw std::runtime_error("long_term_planner (MI355X): stride < 1");
    sample_stride_ = stride; markDirty();
  }

  /** @brief NEW, off by default: reject a q_goal outside [q_min, q_max] before planning (LTP_STATUS_GOAL_OUTSIDE;
   * planTrajectory then returns false with traj untouched). The reference leaves q_goal unchecked (cc:68-77). */
  inline void setGoalCheck(bool enabled) { goal_check_ = enabled; markDirty(); }

  /** @brief NEW, default false: follow the MATLAB original LTPlanner.m where the C++ translation diverges
   * from it (positional root picks, zeros instead of failures, no position limits, 1-based sampler; ltp_hip.h
   * LTP_SEMANTICS_MATLAB). BatchTrajectory::status may then carry LTP_STATUS_MATLAB_ERROR / LTP_STATUS_MATLAB_COMPLEX. */
  inline void setMatlabSemantics(bool enabled) { semantics_ = enabled ? LTP_SEMANTICS_MATLAB : LTP_SEMANTICS_CPP; markDirty(); }

  /** @brief NEW, default TRUE: the reference's pow(x, 3 | 4 | 6) and pow(x, 1.0 / 2) calls (cc:125-331, 378-621) are formed as glibc's
   * pow forms them, operation for operation: every switching time and every sample has the bits of the reference built with gcc +
   * glibc (>= 2.28) on a host with FMA (ltp_hip.h LTP_POW_LIBM). false: the correctly rounded powers instead (LTP_POW_EXACT: within
   * 1 ulp of any libm, switching times within 5e-11 s of the above). powRuleMatchingHostLibm() tells which one this host's libm is. */
  inline void setLibmPow(bool enabled) { pow_rule_ = enabled ? LTP_POW_LIBM : LTP_POW_EXACT; markDirty(); }

  /** @brief NEW: which pow rule reproduces the libm of THIS process (the one a reference built on this host calls): LTP_POW_LIBM
   * (the default rule gives that reference's records bit for bit), LTP_POW_EXACT (setLibmPow(false) does), or -1 — a third libm:
   * about one power in a thousand differs in its last bit from either rule, i.e. ~2 plans per million carry a jerk sample beyond
   * 1e-9 of that reference whichever rule is set. Host only, ~20 ms, no GPU work (ltp_hip.h ltp_host_libm_pow_rule). */
  static inline int powRuleMatchingHostLibm() { return ltp_host_libm_pow_rule(0, nullptr, nullptr); }

  /** @brief NEW, default TRUE since round 6: the envelope calls evaluate only the samples at the ends of each run
   * stretch and either side of the real roots of q'(m) instead of every sample (ltp_hip.h LTP_ENVELOPE_ANALYTIC): identical to the
   * exhaustive result in 8.8e9 soaked window values, <= 1e-12 by construction. false: every sample (the rows' bits by construction). */
  inline void setAnalyticEnvelopes(bool enabled) { envelope_mode_ = enabled ? LTP_ENVELOPE_ANALYTIC : LTP_ENVELOPE_EXHAUSTIVE; markDirty(); }

  /** @brief NEW: HIP device ordinal used by this planner (default 0). */
  inline void setDevice(int device) { if (device != device_) { device_ = device; markDirty(); } }

  /** @brief NEW: the C-ABI handle, for callers that drive the device-pointer entry points of ltp_hip.h directly. */
  ltp_planner* nativeHandle() { return handle(); }

 protected:
  /** @brief reference long_term_planner.h:223-231, cc:82-353 */
  bool optSwitchTimes(int joint, double q_goal, double q_0, double v_0, double a_0, double v_drive, std::array<double, 7>& t,
                      double& dir, char& mod_jerk_profile) {
    int ok = 0;
    ltp_planner* h = handle();
    const int rc = ltp_opt_switch_times_host(h, joint, q_goal, q_0, v_0, a_0, v_drive, t.data(), &dir, &mod_jerk_profile, &ok);
    if (rc != LTP_OK) raise(h, rc, "ltp_opt_switch_times_host");
    return ok != 0;
  }

  /** @brief reference long_term_planner.h:249-259, cc:358-645 */
  bool timeScaling(int joint, double q_goal, double q_0, double v_0, double a_0, double dir, double t_required,
                   std::array<double, 7>& scaled_t, double& v_drive, char& mod_jerk_profile) {
    int ok = 0;
    ltp_planner* h = handle();
    const int rc = ltp_time_scaling_host(h, joint, q_goal, q_0, v_0, a_0, dir, t_required, scaled_t.data(), &v_drive,
                                         &mod_jerk_profile, &ok, nullptr);
    if (rc != LTP_OK) raise(h, rc, "ltp_time_scaling_host");
    return ok != 0;
  }

  /** @brief reference long_term_planner.h:279-285, cc:650-701 (writes t_rel[0..2] only) */
  bool optBraking(int joint, double v_0, double a_0, double& q, std::array<double, 7>& t_rel, double& dir) {
    ltp_planner* h = handle();
    const int rc = ltp_opt_braking_host(h, joint, v_0, a_0, &q, t_rel.data(), &dir);
    if (rc != LTP_OK) raise(h, rc, "ltp_opt_braking_host");
    return true;
  }

  /** @brief reference long_term_planner.h:299-307, cc:706-841 */
  Trajectory getTrajectory(const std::vector<std::array<double, 7>>& t, const std::vector<double>& dir,
                           const std::vector<char>& mod_jerk_profile, const std::vector<double>& q_0,
                           const std::vector<double>& v_0, const std::vector<double>& a_0, const std::vector<double>& v_drive) {
    ltp_planner* h = handle();
    BatchTrajectory b;
    b.n = 1; b.dof = dof_; b.t_sample = t_sample_;
    b.length.assign(1, 0); b.status.assign(1, 0); b.offsets.assign(2, 0ull); b.stored.assign(1, 0);
    std::vector<signed char> mod(mod_jerk_profile.begin(), mod_jerk_profile.end());
    double* packed = nullptr;
    const int rc = ltp_get_trajectory_host(h, 1, t.empty() ? nullptr : t[0].data(), dir.data(), mod.data(), q_0.data(), v_0.data(),
                                           a_0.data(), v_drive.data(), b.length.data(), b.status.data(), b.offsets.data(), &packed);
    if (rc != LTP_OK) raise(h, rc, "ltp_get_trajectory_host");
    if (packed) {
      b.packed.assign(packed, packed + b.offsets[1]);
      ltp_free_host(packed);
    }
    b.stored[0] = ltp_stored_samples(h, b.length[0]);
    return b.trajectory(0);
  }
};
}  // namespace long_term_planner

#endif  // long_term_planner_H
