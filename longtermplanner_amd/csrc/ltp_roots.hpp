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

// scalar traits of the iteration: binary64 for the planner, binary32 only for the reference's float known-answer test
// (tests/src/roots_tests.cc:9-32, through long_term_planner/roots.h)
template <typename R> struct RealTraits;
template <> struct RealTraits<double> {
    static LTP_DEV double min() { return kDblMin; }
    static LTP_DEV double eps() { return kDblEps; }
    static LTP_DEV double sqrt(double x) { return __builtin_sqrt(x); }
    static LTP_DEV double inf() { return __builtin_huge_val(); }
};
template <> struct RealTraits<float> {
    static LTP_DEV float min() { return 1.17549435e-38f; }
    static LTP_DEV float eps() { return 1.1920929e-07f; }
    static LTP_DEV float sqrt(float x) { return __builtin_sqrtf(x); }
    static LTP_DEV float inf() { return __builtin_huge_valf(); }
};
template <typename R> LTP_DEV R rabs(R x) { return x < R(0) ? -x : x; }
template <> LTP_DEV double rabs<double>(double x) { return __builtin_fabs(x); }
template <typename R> LTP_DEV R rmax(R a, R b) { return a < b ? b : a; }
template <typename R> LTP_DEV bool rfinite(R x) { return rabs(x) < RealTraits<R>::inf(); }

// Eigen MatrixBase::makeHouseholder on (w0; w1, w2)
template <typename R>
LTP_DEV void householder3(R w0, R w1, R w2, R& e0, R& e1, R& tau, R& beta)
{
    R tail_sq = w1 * w1 + w2 * w2;
    if (tail_sq <= RealTraits<R>::min()) {
        tau = R(0); beta = w0; e0 = R(0); e1 = R(0);
    } else {
        R b = RealTraits<R>::sqrt(w0 * w0 + tail_sq);
        if (w0 >= R(0)) b = -b;
        e0 = w1 / (w0 - b);
        e1 = w2 / (w0 - b);
        tau = (b - w0) / b;
        beta = b;
    }
}

template <typename R>
LTP_DEV void householder2(R w0, R w1, R& e0, R& tau, R& beta)
{
    R tail_sq = w1 * w1;
    if (tail_sq <= RealTraits<R>::min()) {
        tau = R(0); beta = w0; e0 = R(0);
    } else {
        R b = RealTraits<R>::sqrt(w0 * w0 + tail_sq);
        if (w0 >= R(0)) b = -b;
        e0 = w1 / (w0 - b);
        tau = (b - w0) / b;
        beta = b;
    }
}

// Eigen JacobiRotation::makeGivens, real case
template <typename R>
LTP_DEV void givens(R p, R q, R& c, R& s)
{
    if (q == R(0)) {
        c = p < R(0) ? R(-1) : R(1);
        s = R(0);
    } else if (p == R(0)) {
        c = R(0);
        s = q < R(0) ? R(1) : R(-1);
    } else if (rabs(p) > rabs(q)) {
        R t = q / p;
        R u = RealTraits<R>::sqrt(R(1) + t * t);
        if (p < R(0)) u = -u;
        c = R(1) / u;
        s = -t * c;
    } else {
        R t = p / q;
        R u = RealTraits<R>::sqrt(R(1) + t * t);
        if (q < R(0)) u = -u;
        s = R(-1) / u;
        c = -t * s;
    }
}


// One Francis double-shift step on the active window rows / columns 0..3 (il == 0, iu == 3), every subscript a compile-time
// constant and no lane predicates except the start row im in {0, 1}: what the general step below does for such a window,
// element for element and in the same order — minus the entries right of column 3, which no eigenvalue ever reads (a step only
// feeds on the window itself: diagonal, sub-diagonal and the three columns / rows of the bulge; the columns beyond iu are the
// coupling block of the Schur form, which Eigen carries along for its eigenvectors). Measured on the polynomials the planner
// solves (tools/schur_iters.py): 92 % of all Francis steps of the degree-6 solves, 77 % of degree 5 and all of degree 4 are
// taken on exactly this window — after the first step the two bottom roots of a degree-6 polynomial split off and the
// remaining 4 x 4 block is what converges slowly (the slowest lane of a 100 k batch: 75 of its 76 steps).
template <int N, typename R>
LTP_DEV void francis_step_window4(R (&T)[N][N], R sh0, R sh1, R sh2)
{
    static_assert(N >= 4, "needs a 4 x 4 window");
    // initFrancisQRStep: m = 1, then m = 0
    int im;
    R v0, v1, v2;
    {
        const R Tmm = T[1][1];
        const R r = sh0 - Tmm, s = sh1 - Tmm;
        v0 = (r * s - sh2) / T[2][1] + T[1][2];
        v1 = T[2][2] - Tmm - r - s;
        v2 = T[3][2];
        im = 1;
        const R lhs = T[1][0] * (rabs(v1) + rabs(v2));
        const R rhs = v0 * (rabs(T[0][0]) + rabs(Tmm) + rabs(T[2][2]));
        if (!(rabs(lhs) < RealTraits<R>::eps() * rhs)) {
            const R T00 = T[0][0];
            const R r0 = sh0 - T00, s0 = sh1 - T00;
            v0 = (r0 * s0 - sh2) / T[1][0] + T[0][1];
            v1 = T[1][1] - T00 - r0 - s0;
            v2 = T[2][1];
            im = 0;
        }
    }
    // performFrancisQRStep, k = 0 (only when the step starts at row 0)
    if (im == 0) {
        R e0, e1, tau, beta;
        householder3(v0, v1, v2, e0, e1, tau, beta);
        if (beta != R(0) && tau != R(0)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                R tmp = e0 * T[1][j] + e1 * T[2][j];
                tmp += T[0][j];
                T[0][j] -= tau * tmp;
                T[1][j] -= (tau * e0) * tmp;
                T[2][j] -= (tau * e1) * tmp;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                R tmp = T[i][1] * e0 + T[i][2] * e1;
                tmp += T[i][0];
                T[i][0] -= tau * tmp;
                T[i][1] -= (tau * tmp) * e0;
                T[i][2] -= (tau * tmp) * e1;
            }
        }
    }
    // k = 1
    {
        const bool first = im == 1;
        R w0, w1, w2;
        if (first) { w0 = v0; w1 = v1; w2 = v2; }
        else { w0 = T[1][0]; w1 = T[2][0]; w2 = T[3][0]; }
        R e0, e1, tau, beta;
        householder3(w0, w1, w2, e0, e1, tau, beta);
        if (beta != R(0)) {
            if (first) T[1][0] = -T[1][0];      // k = 1 > il = 0
            else T[1][0] = beta;
            if (tau != R(0)) {
#pragma unroll
                for (int j = 1; j < 4; ++j) {
                    R tmp = e0 * T[2][j] + e1 * T[3][j];
                    tmp += T[1][j];
                    T[1][j] -= tau * tmp;
                    T[2][j] -= (tau * e0) * tmp;
                    T[3][j] -= (tau * e1) * tmp;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    R tmp = T[i][2] * e0 + T[i][3] * e1;
                    tmp += T[i][1];
                    T[i][1] -= tau * tmp;
                    T[i][2] -= (tau * tmp) * e0;
                    T[i][3] -= (tau * tmp) * e1;
                }
            }
        }
    }
    // last 2-vector reflector at (2, 1)
    {
        R e0, tau, beta;
        householder2(T[2][1], T[3][1], e0, tau, beta);
        if (beta != R(0)) {
            T[2][1] = beta;
            if (tau != R(0)) {
#pragma unroll
                for (int j = 2; j < 4; ++j) {
                    R tmp = e0 * T[3][j];
                    tmp += T[2][j];
                    T[2][j] -= tau * tmp;
                    T[3][j] -= (tau * e0) * tmp;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    R tmp = T[i][3] * e0;
                    tmp += T[i][2];
                    T[i][2] -= tau * tmp;
                    T[i][3] -= (tau * tmp) * e0;
                }
            }
        }
    }
    // clean up pollution due to round-off errors
    if (im == 0) { T[2][0] = R(0); T[3][0] = R(0); }
    T[3][1] = R(0);
}

// RealSchur::compute on the monic companion matrix of p (highest coefficient first): T is left quasi-triangular, in the
// scaled units (multiply by scale_out). false: non-finite matrix or no convergence within 40 N iterations.
template <int N, typename R>
__device__ bool real_schur_companion(const R (&p)[N + 1], R (&T)[N][N], R& scale_out)
{
    static_assert(N >= 1 && N <= 8, "degree out of range");
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) T[i][j] = R(0);
#pragma unroll
    for (int i = 1; i < N; ++i) T[i][i - 1] = R(1);
    bool finite = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        R c = (R(-1) * p[N - i]) / p[0];
        T[i][N - 1] = c;
        finite = finite && rfinite(c);
    }
    if (!finite) return false;

    // RealSchur::compute: scale to max|a_ij| == 1 (the sub-diagonal ones make scale >= 1)
    R scale = R(0);
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) scale = rmax(scale, rabs(T[i][j]));
    if constexpr (N == 1) {
        // Eigen: a matrix with max|a_ij| < the smallest normal number is "considered zero" — T = 0, Success. Only degree 1
        // can get here (a x + 0): from degree 2 on the sub-diagonal ones make scale >= 1.
        if (scale < RealTraits<R>::min()) {
            T[0][0] = R(0);
            scale_out = R(1);
            return true;
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) T[i][j] = T[i][j] / scale;

    R norm = R(0);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        R colsum = R(0);
#pragma unroll
        for (int i = 0; i < N; ++i)
            if (i < j + 2) colsum += rabs(T[i][j]);
        norm += colsum;
    }
    const R consider_zero = rmax(norm * (RealTraits<R>::eps() * RealTraits<R>::eps()), RealTraits<R>::min());

    int iu = N - 1, iter = 0, total_iter = 0;
    const int max_iters = 40 * N;
    R exshift = R(0);
    bool converged = true;

    while (iu >= 0) {
        const unsigned long long wave_active = N >= 5 ? __builtin_amdgcn_ballot_w64(true) : 0ull;   // the lanes of the wave still iterating
        // findSmallSubdiagEntry
        int il = iu;
        {
            bool stop = false;
#pragma unroll
            for (int r = N - 1; r >= 1; --r) {
                if (r <= iu && !stop) {
                    R s = rabs(T[r - 1][r - 1]) + rabs(T[r][r]);
                    s = rmax(s * RealTraits<R>::eps(), consider_zero);
                    if (rabs(T[r][r - 1]) <= s) stop = true;
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
                    if (u > 0) T[u][u > 0 ? u - 1 : 0] = R(0);
                }
            }
            iu -= 1;
            iter = 0;
        } else if (il == iu - 1) {
            // two roots found: splitOffTwoRows
#pragma unroll
            for (int u = 1; u < N; ++u) {
                if (u == iu) {
                    R pp = R(0.5) * (T[u - 1][u - 1] - T[u][u]);
                    R qq = pp * pp + T[u][u - 1] * T[u - 1][u];
                    T[u][u] += exshift;
                    T[u - 1][u - 1] += exshift;
                    if (qq >= R(0)) {
                        R z = RealTraits<R>::sqrt(rabs(qq));
                        R c, s;
                        if (pp >= R(0)) givens(pp + z, T[u][u - 1], c, s);
                        else givens(pp - z, T[u][u - 1], c, s);
#pragma unroll
                        for (int k = u - 1; k < N; ++k) {
                            R x = T[u - 1][k], y = T[u][k];
                            T[u - 1][k] = c * x + (-s) * y;
                            T[u][k] = s * x + c * y;
                        }
#pragma unroll
                        for (int k = 0; k <= u; ++k) {
                            R x = T[k][u - 1], y = T[k][u];
                            T[k][u - 1] = c * x + (-s) * y;
                            T[k][u] = s * x + c * y;
                        }
                        T[u][u - 1] = R(0);
                    }
                    if (u > 1) T[u - 1][u > 1 ? u - 2 : 0] = R(0);
                }
            }
            iu -= 2;
            iter = 0;
        } else {
            // one Francis double-shift step on the window [il, iu], iu >= il + 2
            R sh0 = R(0), sh1 = R(0), sh2 = R(0), sub0 = R(0), sub1 = R(0);
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
                R s = rabs(sub0) + rabs(sub1);
                sh0 = R(0.75) * s;
                sh1 = R(0.75) * s;
                sh2 = R(-0.4375) * s * s;
            }
            if (iter == 30) {
                R s = (sh1 - sh0) / R(2);
                s = s * s + sh2;
                if (s > R(0)) {
                    s = RealTraits<R>::sqrt(s);
                    if (sh1 < sh0) s = -s;
                    s = s + (sh1 - sh0) / R(2);
                    s = sh0 - sh2 / s;
                    exshift += s;
#pragma unroll
                    for (int i = 0; i < N; ++i)
                        if (i <= iu) T[i][i] -= s;
                    sh0 = sh1 = sh2 = R(0.964);
                }
            }
            iter += 1;
            total_iter += 1;
            if (total_iter > max_iters) { converged = false; break; }

            if constexpr (N >= 5) {
                // every lane that takes a step now is on the window rows 0..3: the specialised step (same arithmetic, far fewer
                // instructions: no lane predicates, no columns beyond the window). Degree 4 keeps the general step: its whole
                // matrix is that window, its solves are 3-5 steps long, and the second code path costs registers in every kernel
                // that carries a quartic site.
                if (__builtin_amdgcn_ballot_w64(!(il == 0 && iu == 3)) == 0ull) {
                    francis_step_window4<N, R>(T, sh0, sh1, sh2);
                    // ... and as long as these lanes are the only ones of the wave still iterating and none of them sees a
                    // negligible sub-diagonal entry, the iterations stay here: findSmallSubdiagEntry, the shifts, the two
                    // exceptional shifts and the step, all written for iu == 3, il == 0 — the same operations on the same
                    // elements, without the general loop's searches over N rows (745 -> ~520 instructions per iteration; the
                    // slowest lane of a batch spends 73 of its 76 iterations in this state). Leaving changes nothing: the general
                    // loop looks at the same three entries again.
                    if (__builtin_amdgcn_ballot_w64(true) == wave_active) {
                        for (;;) {
                            bool small = false;
                            {
                                R s = rabs(T[2][2]) + rabs(T[3][3]);
                                s = rmax(s * RealTraits<R>::eps(), consider_zero);
                                if (rabs(T[3][2]) <= s) small = true;
                                else {
                                    s = rabs(T[1][1]) + rabs(T[2][2]);
                                    s = rmax(s * RealTraits<R>::eps(), consider_zero);
                                    if (rabs(T[2][1]) <= s) small = true;
                                    else {
                                        s = rabs(T[0][0]) + rabs(T[1][1]);
                                        s = rmax(s * RealTraits<R>::eps(), consider_zero);
                                        if (rabs(T[1][0]) <= s) small = true;
                                    }
                                }
                            }
                            if (__builtin_amdgcn_ballot_w64(small) != 0ull) break;
                            R h0 = T[3][3], h1 = T[2][2], h2 = T[3][2] * T[2][3];
                            if (iter == 10) {
                                exshift += h0;
                                T[0][0] -= h0; T[1][1] -= h0; T[2][2] -= h0; T[3][3] -= h0;
                                R s = rabs(T[3][2]) + rabs(T[2][1]);
                                h0 = R(0.75) * s;
                                h1 = R(0.75) * s;
                                h2 = R(-0.4375) * s * s;
                            }
                            if (iter == 30) {
                                R s = (h1 - h0) / R(2);
                                s = s * s + h2;
                                if (s > R(0)) {
                                    s = RealTraits<R>::sqrt(s);
                                    if (h1 < h0) s = -s;
                                    s = s + (h1 - h0) / R(2);
                                    s = h0 - h2 / s;
                                    exshift += s;
                                    T[0][0] -= s; T[1][1] -= s; T[2][2] -= s; T[3][3] -= s;
                                    h0 = h1 = h2 = R(0.964);
                                }
                            }
                            iter += 1;
                            total_iter += 1;
                            if (total_iter > max_iters) { converged = false; break; }
                            francis_step_window4<N, R>(T, h0, h1, h2);
                        }
                        if (!converged) break;
                    }
                    continue;
                }
            }

            // initFrancisQRStep
            int im = il;
            R v0 = R(0), v1 = R(0), v2 = R(0);
            {
                bool found = false;
#pragma unroll
                for (int m = N - 3; m >= 0; --m) {
                    if (m <= iu - 2 && m >= il && !found) {
                        const R Tmm = T[m][m];
                        const R r = sh0 - Tmm;
                        const R s = sh1 - Tmm;
                        v0 = (r * s - sh2) / T[m + 1][m] + T[m][m + 1];
                        v1 = T[m + 1][m + 1] - Tmm - r - s;
                        v2 = T[m + 2][m + 1];
                        im = m;
                        if (m == il) {
                            found = true;
                        } else {
                            const int mm1 = m > 0 ? m - 1 : 0;
                            const R lhs = T[m][mm1] * (rabs(v1) + rabs(v2));
                            const R rhs = v0 * (rabs(T[mm1][mm1]) + rabs(Tmm) + rabs(T[m + 1][m + 1]));
                            if (rabs(lhs) < RealTraits<R>::eps() * rhs) found = true;
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
                    R w0, w1, w2;
                    if (first) { w0 = v0; w1 = v1; w2 = v2; }
                    else { w0 = T[k][km1]; w1 = T[k + 1][km1]; w2 = T[k + 2][km1]; }
                    R e0, e1, tau, beta;
                    householder3(w0, w1, w2, e0, e1, tau, beta);
                    if (beta != R(0)) {
                        if (first && k > il) T[k][km1] = -T[k][km1];
                        else if (!first) T[k][km1] = beta;
                        if (tau != R(0)) {
#pragma unroll
                            for (int j = k; j < N; ++j) {
                                R tmp = e0 * T[k + 1][j] + e1 * T[k + 2][j];
                                tmp += T[k][j];
                                T[k][j] -= tau * tmp;
                                T[k + 1][j] -= (tau * e0) * tmp;
                                T[k + 2][j] -= (tau * e1) * tmp;
                            }
                            const int rmax = iu < k + 3 ? iu : k + 3;
#pragma unroll
                            for (int i = 0; i < N; ++i) {
                                if (i <= k + 3 && i <= rmax) {
                                    R tmp = T[i][k + 1] * e0 + T[i][k + 2] * e1;
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
                    R e0, tau, beta;
                    householder2(T[u - 1][u - 2], T[u][u - 2], e0, tau, beta);
                    if (beta != R(0)) {
                        T[u - 1][u - 2] = beta;
                        if (tau != R(0)) {
#pragma unroll
                            for (int j = u - 1; j < N; ++j) {
                                R tmp = e0 * T[u][j];
                                tmp += T[u - 1][j];
                                T[u - 1][j] -= tau * tmp;
                                T[u][j] -= (tau * e0) * tmp;
                            }
#pragma unroll
                            for (int i = 0; i <= u; ++i) {
                                R tmp = T[i][u] * e0;
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
                    T[i][i - 2] = R(0);
                    if (i > im + 2) T[i][i >= 3 ? i - 3 : 0] = R(0);
                }
            }
        }
    }
    scale_out = scale;
    return converged;
}

// Smallest admissible root of p[0] x^N + ... + p[N] (highest coefficient first).
template <int N>
__device__ double smallest_positive_real_root(const double (&p)[N + 1])
{
    static_assert(N >= 3 && N <= 8, "degree out of range");
    double T[N][N];
    double scale;
    const bool converged = real_schur_companion<N, double>(p, T, scale);
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

// All eigenvalues of the companion matrix, in the order and with the conjugate-pair convention of Eigen 3.4's
// EigenSolver::compute (roots.h:32-33): read off the quasi-triangular T top to bottom; a 1x1 block is a real eigenvalue
// (imaginary part exactly zero), a 2x2 block that splitOffTwoRows could not split a pair (re, +im), (re, -im).
// A non-finite matrix or a non-converged iteration yields NaN everywhere (Eigen leaves the vector uninitialised).
template <int N, typename R>
__device__ void companion_eigenvalues(const R (&p)[N + 1], R (&re)[N], R (&im)[N])
{
    R T[N][N];
    R scale;
    const bool ok = real_schur_companion<N, R>(p, T, scale);
    const R nan = RealTraits<R>::inf() - RealTraits<R>::inf();
    if (!ok) {
#pragma unroll
        for (int i = 0; i < N; ++i) { re[i] = nan; im[i] = nan; }
        return;
    }
    bool skip = false;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (skip) { skip = false; continue; }
        const R tii = T[i][i] * scale;
        bool is_real = true;
        R sub = R(0);
        if (i < N - 1) {
            sub = T[i + 1 < N ? i + 1 : i][i] * scale;
            is_real = (sub == R(0));
        }
        if (is_real) {
            re[i] = tii;
            im[i] = R(0);
        } else {
            const int ip = i + 1 < N ? i + 1 : i;
            const R tpp = T[ip][ip] * scale;
            const R pp = R(0.5) * (tii - tpp);
            R t0 = sub, t1 = T[i][ip] * scale;
            const R maxval = rmax(rabs(pp), rmax(rabs(t0), rabs(t1)));
            t0 /= maxval;
            t1 /= maxval;
            const R p0 = pp / maxval;
            const R z = maxval * RealTraits<R>::sqrt(rabs(p0 * p0 + t0 * t1));
            re[i] = tpp + pp; im[i] = z;
            re[ip] = tpp + pp; im[ip] = -z;
            skip = true;
        }
    }
}

}  // namespace ltp
