// long_term_planner/roots.h — drop-in for the reference header of the same path, without Eigen.
//
// The reference's roots.h (/root/reference/include/long_term_planner/roots.h:22-50) builds the monic companion matrix of a
// polynomial and asks Eigen 3.4's EigenSolver for its eigenvalues; getSmallestPositiveNonComplexRoot then picks the
// smallest eigenvalue with imag == 0 exactly and real > 1e-7. Here the eigen-solve runs on the MI355X (ltp_roots_f64_host /
// ltp_roots_f32_host of include/ltp_hip.h: the same published RealSchur / hqr2 iteration, so the eigenvalues come in
// Eigen's output order with its conjugate-pair convention), and — Eigen being neither vendored by the reference nor part
// of this library — the Eigen matrix types of the two signatures become std::vector:
//   reference   Eigen::Matrix<std::complex<T>, Dynamic, Dynamic> roots(Eigen::Matrix<T, Dynamic, 1> poly_vals)
//   here        std::vector<std::complex<T>>                      roots(const std::vector<T>& poly_vals)      T = float, double
// Degrees 1..8. A HIP device is required (std::runtime_error otherwise: there is no CPU fallback).
//
// For users who DO have Eigen: when <Eigen/Dense> is on the include path (and LTP_ROOTS_NO_EIGEN is not defined) the reference's
// exact signatures exist as well — roots<T>(Eigen::Matrix<T, Dynamic, 1>) returning an n x 1 complex Eigen matrix and
// getSmallestPositiveNonComplexRoot<T>(Eigen::Matrix<std::complex<T>, Dynamic, Dynamic>) — delegating to the same device call, so
// the reference's tests/src/roots_tests.cc:9-32 compiles against this header unchanged. Eigen is not installed in this repository's
// build image: the block is compiled there against tests/cpp/not_eigen — a container-only stand-in for the type names, NOT Eigen —
// which proves syntax and overload resolution (tests/cpp/roots_eigen_signature_test.cc) and lets the reference's roots_tests.cc run
// against the device (tests/test_gpu_kat.py); against the real Eigen headers it is untested.
#ifndef roots_H
#define roots_H

#include <cmath>
#include <complex>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "ltp_hip.h"

namespace long_term_planner {

namespace detail {
// one configuration-free handle (dof 0) on device 0 for the eigen-solves of this header
inline ltp_planner* rootsHandle() {
  static std::once_flag once;
  static ltp_planner* handle = nullptr;
  static int rc = LTP_OK;
  std::call_once(once, [] { rc = ltp_create(0, 0.001, nullptr, nullptr, nullptr, nullptr, nullptr, 0, &handle); });
  if (rc != LTP_OK || !handle)
    throw std::runtime_error("long_term_planner (MI355X): roots() needs a HIP device (ltp_create failed with code " + std::to_string(rc) + ")");
  return handle;
}
inline int rootsCall(ltp_planner* h, int degree, const double* c, double* re, double* im) { return ltp_roots_f64_host(h, 1, degree, c, re, im); }
inline int rootsCall(ltp_planner* h, int degree, const float* c, float* re, float* im) { return ltp_roots_f32_host(h, 1, degree, c, re, im); }
}  // namespace detail

/**
 * @brief All roots of a polynomial: eigenvalues of its companion matrix (reference roots.h:22-34).
 * @param poly_vals polynomial coefficients starting with the highest exponent.
 * @return the roots in the order Eigen 3.4's EigenSolver returns them.
 */
template <class T>
std::vector<std::complex<T>> roots(const std::vector<T>& poly_vals) {
  const int degree = static_cast<int>(poly_vals.size()) - 1;
  if (degree < 1 || degree > 8) throw std::runtime_error("long_term_planner (MI355X): roots() supports degrees 1..8");
  std::vector<T> re(degree), im(degree);
  ltp_planner* h = detail::rootsHandle();
  const int rc = detail::rootsCall(h, degree, poly_vals.data(), re.data(), im.data());
  if (rc != LTP_OK) throw std::runtime_error(std::string("long_term_planner (MI355X): roots() failed: ") + ltp_last_error(h));
  std::vector<std::complex<T>> r(degree);
  for (int i = 0; i < degree; ++i) r[i] = std::complex<T>(re[i], im[i]);
  return r;
}

/** @brief Smallest root with imag == 0 exactly and real > 1e-7, else +INFINITY (reference roots.h:43-50). */
template <class T>
T getSmallestPositiveNonComplexRoot(const std::vector<std::complex<T>>& r) {
  T smallest_val = INFINITY;
  for (const auto& root : r) {
    if (root.imag() == 0 && root.real() > 1e-7) smallest_val = std::min(smallest_val, root.real());
  }
  return smallest_val;
}

}  // namespace long_term_planner

#if !defined(LTP_ROOTS_NO_EIGEN) && defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
namespace long_term_planner {

/** @brief reference roots.h:22-34 with its exact types: the eigenvalues as an n x 1 complex matrix, in Eigen's order. */
template <class T>
Eigen::Matrix<std::complex<T>, Eigen::Dynamic, Eigen::Dynamic> roots(Eigen::Matrix<T, Eigen::Dynamic, 1> poly_vals) {
  std::vector<T> c(static_cast<size_t>(poly_vals.size()));
  for (size_t i = 0; i < c.size(); ++i) c[i] = poly_vals[static_cast<Eigen::Index>(i)];
  const std::vector<std::complex<T>> r = roots<T>(c);
  Eigen::Matrix<std::complex<T>, Eigen::Dynamic, Eigen::Dynamic> out(static_cast<Eigen::Index>(r.size()), 1);
  for (size_t i = 0; i < r.size(); ++i) out(static_cast<Eigen::Index>(i), 0) = r[i];
  return out;
}

/** @brief reference roots.h:43-50 with its exact type (the first column holds the roots). */
template <class T>
T getSmallestPositiveNonComplexRoot(Eigen::Matrix<std::complex<T>, Eigen::Dynamic, Eigen::Dynamic> r) {
  T smallest_val = INFINITY;
  for (Eigen::Index i = 0; i < r.rows(); ++i) {
    const std::complex<T> root = r(i, 0);
    if (root.imag() == 0 && root.real() > 1e-7) smallest_val = std::min(smallest_val, root.real());
  }
  return smallest_val;
}

}  // namespace long_term_planner
#endif
#endif
#endif  // roots_H
