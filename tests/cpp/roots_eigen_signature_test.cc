// roots_eigen_signature_test.cc — the Eigen-typed overloads of include/long_term_planner/roots.h (the reference's exact signatures,
// /root/reference/include/long_term_planner/roots.h:22-23, 43), compiled against tests/cpp/not_eigen (a container-only stand-in:
// NOT Eigen, pins nothing about Eigen) to prove that they compile and that overload resolution picks them for the reference's call
// patterns: roots<float>(Eigen::VectorXf) with the result taken as Matrix<complex<float>, Dynamic, Dynamic> (roots_tests.cc:21-22), and
// getSmallestPositiveNonComplexRoot(result) WITHOUT an explicit template argument (src/long_term_planner.cc:625-626).
// Compiled with -fsyntax-only in the CPU suite; built and RUN on the GPU box, where it also checks the reference's float degree-6
// known-answer table (roots_tests.cc:10-31) and the double form through these overloads.
#include <cmath>
#include <cstdio>
#include <type_traits>

#include <Eigen/Dense>
#include "long_term_planner/roots.h"

using long_term_planner::getSmallestPositiveNonComplexRoot;
using long_term_planner::roots;

typedef Eigen::Matrix<std::complex<float>, Eigen::Dynamic, Eigen::Dynamic> CMatF;
typedef Eigen::Matrix<std::complex<double>, Eigen::Dynamic, Eigen::Dynamic> CMatD;

// overload resolution, checked by the compiler
static_assert(std::is_same<decltype(roots<float>(std::declval<Eigen::VectorXf>())), CMatF>::value, "roots<float>(VectorXf) is the Eigen-typed overload");
static_assert(std::is_same<decltype(roots<double>(std::declval<Eigen::VectorXd>())), CMatD>::value, "roots<double>(VectorXd) is the Eigen-typed overload");
static_assert(std::is_same<decltype(getSmallestPositiveNonComplexRoot(std::declval<CMatF>())), float>::value, "T deduced from the matrix type");
static_assert(std::is_same<decltype(getSmallestPositiveNonComplexRoot(std::declval<CMatD>())), double>::value, "T deduced from the matrix type");
static_assert(std::is_same<decltype(roots<double>(std::declval<std::vector<double>>())), std::vector<std::complex<double>>>::value,
              "the std::vector form is still there");

int main()
{
    // the table of the reference's tests/src/roots_tests.cc:10-31 (data), through the Eigen-typed signatures
    Eigen::VectorXf poly_vals(7);
    poly_vals << 144.f, -1008.f, 2448.f, 3024.0192f, -15768.1344f, 0.f, 22752.40320128f;
    CMatF r = roots<float>(poly_vals);
    const double re[6] = {-1.67276, -1.35687, 2.00001, 2.09261, 2.9685, 2.9685}, im[6] = {0, 0, 0, 0, 2.79663, -2.79663};
    int bad = 0;
    if (r.rows() != 6 || r.cols() != 1) ++bad;
    for (int i = 0; i < 6 && !bad; ++i) {
        if (std::fabs(r(i, 0).real() - re[i]) > 1e-5 || std::fabs(r(i, 0).imag() - im[i]) > (im[i] == 0 ? 1e-9 : 1e-5)) {
            std::printf("root %d: (%g, %g), expected (%g, %g)\n", i, (double)r(i, 0).real(), (double)r(i, 0).imag(), re[i], im[i]);
            ++bad;
        }
    }
    const float smallest = getSmallestPositiveNonComplexRoot(r);                       // T deduced, as at cc:625-626
    if (std::fabs(smallest - 2.00001f) > 1e-5f) { std::printf("smallest positive real root %g\n", (double)smallest); ++bad; }
    Eigen::VectorXd pd(7);
    pd << 144., -1008., 2448., 3024.0192, -15768.1344, 0., 22752.40320128;
    // double: the Eigen-typed pair gives what the std::vector pair gives (one device call behind both)
    const double sd = getSmallestPositiveNonComplexRoot(roots<double>(pd));
    const double sv = getSmallestPositiveNonComplexRoot(roots<double>(std::vector<double>{144., -1008., 2448., 3024.0192, -15768.1344, 0., 22752.40320128}));
    if (!(sd == sv)) { std::printf("double: Eigen-typed %g, std::vector %g\n", sd, sv); ++bad; }
    std::printf("roots_eigen_signature_test: %s\n", bad ? "FAILED" : "ok");
    return bad ? 1 : 0;
}
