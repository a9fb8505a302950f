// ltp_roots.hpp — per-lane polynomial root selection for degrees 4, 5 and 6.
//
// Replaces, on the device, reference include/long_term_planner/roots.h:22-34
// (roots<T>: eigenvalues of the monic companion matrix through Eigen 3.4's
// EigenSolver) and roots.h:43-50 (getSmallestPositiveNonComplexRoot: the smallest
// eigenvalue with imag == 0 exactly and real > 1e-7, else +INFINITY).
//
// MI355X design: one lane solves one polynomial. The N x N Hessenberg matrix lives
// entirely in VGPRs: every loop over matrix indices is fully unrolled so all
// subscripts are compile-time constants, and the data-dependent window of the
// Francis iteration (il, im, iu) is expressed as per-lane predicates. There is no
// scratch memory, no LDS and no cross-lane traffic; divergent lanes simply sit out
// steps their window does not cover. The iteration itself is the published
// EISPACK hqr2 / Eigen RealSchur double-shift QR (scaling by max|a_ij|, Wilkinson's
// ad-hoc shift at iteration 10, MATLAB's at 30, at most 40*N steps), which is what
// decides whether a close pair of roots comes out real (split 2x2 block) or complex.
//
// Defined where the reference is not: a non-finite
// companion matrix or a non-converged iteration selects no root (+INFINITY).
#pragma once
#include "ltp_math.hpp"

namespace ltp {

constexpr double kDblMin = 2.2250738585072014e-308;
constexpr double kDblEps = 2.220446049250313e-16;

// Eigen MatrixBase::makeHouseholder on (w0; w1, w2)
LTP_DEV void householder3(double w0, double w1, double w2, double& e0, double& e1, double& tau, double& beta)
{
    double tail_sq = w1 * w1 + w2 * w2;
    if (tail_sq <= kDblMin) {
        tau = 0.0; beta = w0; e0 = 0.0; e1 = 0.0;
    } else {
        double b = dsqrt(w0 * w0 + tail_sq);
        if (w0 >= 0.0) b = -b;
        e0 = w1 / (w0 - b);
        e1 = w2 / (w0 - b);
        tau = (b - w0) / b;
        beta = b;
    }
}

LTP_DEV void householder2(double w0, double w1, double& e0, double& tau, double& beta)
{
    double tail_sq = w1 * w1;
    if (tail_sq <= kDblMin) {
        tau = 0.0; beta = w0; e0 = 0.0;
    } else {
        double b = dsqrt(w0 * w0 + tail_sq);
        if (w0 >= 0.0) b = -b;
        e0 = w1 / (w0 - b);
        tau = (b - w0) / b;
        beta = b;
    }
}

// Eigen JacobiRotation::makeGivens, real case
LTP_DEV void givens(double p, double q, double& c, double& s)
{
    if (q == 0.0) {
        c = p < 0.0 ? -1.0 : 1.0;
        s = 0.0;
    } else if (p == 0.0) {
        c = 0.0;
        s = q < 0.0 ? 1.0 : -1.0;
    } else if (dabs(p) > dabs(q)) {
        double t = q / p;
        double u = dsqrt(1.0 + t * t);
        if (p < 0.0) u = -u;
        c = 1.0 / u;
        s = -t * c;
    } else {
        double t = p / q;
        double u = dsqrt(1.0 + t * t);
        if (q < 0.0) u = -u;
        s = -1.0 / u;
        c = -t * s;
    }
}

// Smallest admissible root of p[0] x^N + ... + p[N] (highest coefficient first).
template <int N>
__device__ double smallest_positive_real_root(const double (&p)[N + 1])
{
    static_assert(N >= 3 && N <= 8, "degree out of range");
    double T[N][N];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) T[i][j] = 0.0;
#pragma unroll
    for (int i = 1; i < N; ++i) T[i][i - 1] = 1.0;
    bool finite = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double c = (-1.0 * p[N - i]) / p[0];
        T[i][N - 1] = c;
        finite = finite && dfinite(c);
    }
    if (!finite) return kInf;

    // RealSchur::compute: scale to max|a_ij| == 1 (the sub-diagonal ones make scale >= 1)
    double scale = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) scale = dmax(scale, dabs(T[i][j]));
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) T[i][j] = T[i][j] / scale;

    double norm = 0.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double colsum = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < j + 2) colsum += dabs(T[i][j]);
        norm += colsum;
    }
    const double consider_zero = dmax(norm * (kDblEps * kDblEps), kDblMin);

    int iu = N - 1, iter = 0, total_iter = 0;
    const int max_iters = 40 * N;
    double exshift = 0.0;
    bool converged = true;

    while (iu >= 0) {
        // findSmallSubdiagEntry
        int il = iu;
        {
            bool stop = false;
#pragma unroll
            for (int r = N - 1; r >= 1; --r) {
                if (r <= iu && !stop) {
                    double s = dabs(T[r - 1][r - 1]) + dabs(T[r][r]);
                    s = dmax(s * kDblEps, consider_zero);
                    if (dabs(T[r][r - 1]) <= s) stop = true;
                    else il = r - 1;
                }
            }
        }
        if (il == iu) {
            // one root found
#pragma unroll
            for (int u = 0; u < N; ++u) {
                if (u == iu) {
                    T[u][u] = T[u][u] + exshift;
                    if (u > 0) T[u][u > 0 ? u - 1 : 0] = 0.0;
                }
            }
            iu -= 1;
            iter = 0;
        } else if (il == iu - 1) {
            // two roots found: splitOffTwoRows
#pragma unroll
            for (int u = 1; u < N; ++u) {
                if (u == iu) {
                    double pp = 0.5 * (T[u - 1][u - 1] - T[u][u]);
                    double qq = pp * pp + T[u][u - 1] * T[u - 1][u];
                    T[u][u] += exshift;
                    T[u - 1][u - 1] += exshift;
                    if (qq >= 0.0) {
                        double z = dsqrt(dabs(qq));
                        double c, s;
                        if (pp >= 0.0) givens(pp + z, T[u][u - 1], c, s);
                        else givens(pp - z, T[u][u - 1], c, s);
#pragma unroll
                        for (int k = u - 1; k < N; ++k) {
                            double x = T[u - 1][k], y = T[u][k];
                            T[u - 1][k] = c * x + (-s) * y;
                            T[u][k] = s * x + c * y;
                        }
#pragma unroll
                        for (int k = 0; k <= u; ++k) {
                            double x = T[k][u - 1], y = T[k][u];
                            T[k][u - 1] = c * x + (-s) * y;
                            T[k][u] = s * x + c * y;
                        }
                        T[u][u - 1] = 0.0;
                    }
                    if (u > 1) T[u - 1][u > 1 ? u - 2 : 0] = 0.0;
                }
            }
            iu -= 2;
            iter = 0;
        } else {
            // one Francis double-shift step on the window [il, iu], iu >= il + 2
            double sh0 = 0.0, sh1 = 0.0, sh2 = 0.0, sub0 = 0.0, sub1 = 0.0;
#pragma unroll
            for (int u = 2; u < N; ++u) {
                if (u == iu) {
                    sh0 = T[u][u];
                    sh1 = T[u - 1][u - 1];
                    sh2 = T[u][u - 1] * T[u - 1][u];
                    sub0 = T[u][u - 1];
                    sub1 = T[u - 1][u - 2];
                }
            }
            if (iter == 10) {
                exshift += sh0;
#pragma unroll
                for (int i = 0; i < N; ++i)
                    if (i <= iu) T[i][i] -= sh0;
                double s = dabs(sub0) + dabs(sub1);
                sh0 = 0.75 * s;
                sh1 = 0.75 * s;
                sh2 = -0.4375 * s * s;
            }
            if (iter == 30) {
                double s = (sh1 - sh0) / 2.0;
                s = s * s + sh2;
                if (s > 0.0) {
                    s = dsqrt(s);
                    if (sh1 < sh0) s = -s;
                    s = s + (sh1 - sh0) / 2.0;
                    s = sh0 - sh2 / s;
                    exshift += s;
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i <= iu) T[i][i] -= s;
                    sh0 = sh1 = sh2 = 0.964;
                }
            }
            iter += 1;
            total_iter += 1;
            if (total_iter > max_iters) { converged = false; break; }

            // initFrancisQRStep
            int im = il;
            double v0 = 0.0, v1 = 0.0, v2 = 0.0;
            {
                bool found = false;
#pragma unroll
                for (int m = N - 3; m >= 0; --m) {
                    if (m <= iu - 2 && m >= il && !found) {
                        const double Tmm = T[m][m];
                        const double r = sh0 - Tmm;
                        const double s = sh1 - Tmm;
                        v0 = (r * s - sh2) / T[m + 1][m] + T[m][m + 1];
                        v1 = T[m + 1][m + 1] - Tmm - r - s;
                        v2 = T[m + 2][m + 1];
                        im = m;
                        if (m == il) {
                            found = true;
                        } else {
                            const int mm1 = m > 0 ? m - 1 : 0;
                            const double lhs = T[m][mm1] * (dabs(v1) + dabs(v2));
                            const double rhs = v0 * (dabs(T[mm1][mm1]) + dabs(Tmm) + dabs(T[m + 1][m + 1]));
                            if (dabs(lhs) < kDblEps * rhs) found = true;
                        }
                    }
                }
            }
            // performFrancisQRStep: chase the bulge from im to iu-2
#pragma unroll
            for (int k = 0; k <= N - 3; ++k) {
                if (k >= im && k <= iu - 2) {
                    const bool first = (k == im);
                    const int km1 = k > 0 ? k - 1 : 0;
                    double w0, w1, w2;
                    if (first) { w0 = v0; w1 = v1; w2 = v2; }
                    else { w0 = T[k][km1]; w1 = T[k + 1][km1]; w2 = T[k + 2][km1]; }
                    double e0, e1, tau, beta;
                    householder3(w0, w1, w2, e0, e1, tau, beta);
                    if (beta != 0.0) {
                        if (first && k > il) T[k][km1] = -T[k][km1];
                        else if (!first) T[k][km1] = beta;
                        if (tau != 0.0) {
#pragma unroll
                            for (int j = k; j < N; ++j) {
                                double tmp = e0 * T[k + 1][j] + e1 * T[k + 2][j];
                                tmp += T[k][j];
                                T[k][j] -= tau * tmp;
                                T[k + 1][j] -= (tau * e0) * tmp;
                                T[k + 2][j] -= (tau * e1) * tmp;
                            }
                            const int rmax = iu < k + 3 ? iu : k + 3;
#pragma unroll
                            for (int i = 0; i < N; ++i) {
                                if (i <= k + 3 && i <= rmax) {
                                    double tmp = T[i][k + 1] * e0 + T[i][k + 2] * e1;
                                    tmp += T[i][k];
                                    T[i][k] -= tau * tmp;
                                    T[i][k + 1] -= (tau * tmp) * e0;
                                    T[i][k + 2] -= (tau * tmp) * e1;
                                }
                            }
                        }
                    }
                }
            }
            // last 2-vector reflector at (iu-1, iu-2)
#pragma unroll
            for (int u = 2; u < N; ++u) {
                if (u == iu) {
                    double e0, tau, beta;
                    householder2(T[u - 1][u - 2], T[u][u - 2], e0, tau, beta);
                    if (beta != 0.0) {
                        T[u - 1][u - 2] = beta;
                        if (tau != 0.0) {
#pragma unroll
                            for (int j = u - 1; j < N; ++j) {
                                double tmp = e0 * T[u][j];
                                tmp += T[u - 1][j];
                                T[u - 1][j] -= tau * tmp;
                                T[u][j] -= (tau * e0) * tmp;
                            }
#pragma unroll
                            for (int i = 0; i <= u; ++i) {
                                double tmp = T[i][u] * e0;
                                tmp += T[i][u - 1];
                                T[i][u - 1] -= tau * tmp;
                                T[i][u] -= (tau * tmp) * e0;
                            }
                        }
                    }
                }
            }
            // clean up pollution due to round-off errors
#pragma unroll
            for (int i = 2; i < N; ++i) {
                if (i >= im + 2 && i <= iu) {
                    T[i][i - 2] = 0.0;
                    if (i > im + 2) T[i][i >= 3 ? i - 3 : 0] = 0.0;
                }
            }
        }
    }
    if (!converged) return kInf;

    // EigenSolver::compute: eigenvalues off the quasi-triangular T (after T *= scale),
    // folded with the selection rule of roots.h:43-50.
    double best = kInf;
    bool skip = false, bad = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (skip) { skip = false; continue; }
        if (bad) continue;
        const double tii = T[i][i] * scale;
        bool is_real = true;
        double sub = 0.0;
        if (i < N - 1) {
            sub = T[i + 1 < N ? i + 1 : i][i] * scale;
            is_real = (sub == 0.0);
        }
        if (is_real) {
            if (!dfinite(tii)) { bad = true; continue; }
            if (tii > 1e-7) best = dmin(best, tii);
        } else {
            const int ip = i + 1 < N ? i + 1 : i;
            const double tpp = T[ip][ip] * scale;
            const double pp = 0.5 * (tii - tpp);
            double t0 = sub, t1 = T[i][ip] * scale;
            const double maxval = dmax(dabs(pp), dmax(dabs(t0), dabs(t1)));
            t0 /= maxval;
            t1 /= maxval;
            const double p0 = pp / maxval;
            const double z = maxval * dsqrt(dabs(p0 * p0 + t0 * t1));
            const double er = tpp + pp;
            if (!dfinite(er) || !dfinite(z)) { bad = true; continue; }
            if (z == 0.0 && er > 1e-7) best = dmin(best, er);
            skip = true;
        }
    }
    return best;
}

}  // namespace ltp
