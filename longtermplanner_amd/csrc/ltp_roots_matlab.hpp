// ltp_roots_matlab.hpp — MATLAB's roots() on the device, for the MATLAB-semantics mode only.
//
// LTPlanner.m picks polynomial roots BY POSITION in the output of roots() (LTPlanner.m:346, 360, 374, 388, 402, 416) or takes
// the first one that passes a filter (:247-250, 272-275), so this mode needs the eigenvalues of the companion matrix in the
// order MATLAB's eig returns them. MATLAB is not part of the reference tree; what is restated here is roots.m's documented
// construction (strip leading / trailing zero coefficients, A = diag(ones(n-1,1),-1), A(1,:) = -c(2:end)/c(1), eig(A)) and
// LAPACK's published DGEEV path for eigenvalues only: DGEBAL('B') (scaling by powers of two; the permutation search finds
// nothing on a companion matrix with a non-zero constant coefficient), DGEHRD (the identity on a Hessenberg matrix), DHSEQR ->
// DLAHQR (double-shift QR with the Ahues-Tisseur deflation test, exceptional shifts every 10 iterations without deflation),
// DLANV2 for the final 2x2 blocks. The order is checked against numpy.roots — the same driver — through the CPU twin of this file
// in the test suite (tests/test_matlab_twin.py), on the HOST against that twin (tests/cpp/matlab_roots_test.cc compiles this very
// header with g++: bit-identical on every class of input) and on the device against it (tests/test_gpu_matlab.py); against
// MATLAB's own LAPACK build it is unpinned.
//
// MI355X design (round 5; rounds 3-4 indexed the matrix dynamically, first in scratch memory, then in 144 KB of dynamic LDS per
// block): one lane solves one polynomial and its N x N matrix lives in VGPRs, as in the C++-semantics solver of ltp_roots.hpp —
// every loop over matrix indices is a compile-time loop (static_for: each subscript a constant), and what LAPACK's loops take
// from the data — the active window [l, i], the start row m of a sweep, whether a reflector spans two or three rows, the order
// n left after stripping zero coefficients — is a per-lane predicate on those constant positions. Entries outside the n x n
// matrix are kept at zero, which makes DGEBAL's norms and DLAHQR's neighbour sums come out as LAPACK forms them without a
// predicate on n. Same operations in the same order on the same elements as the loops of the twin: same bits. No LDS, no
// scratch, no launch convention.
#pragma once
#if defined(__HIPCC__)
#include "ltp_math.hpp"
#define LTP_MR_FN LTP_DEV
// the solver of one degree is ONE out-of-line function per kernel image: a stage kernel has up to six sites that solve a polynomial
// and ~100 live registers of its own around each; inlined, the 36 matrix registers and the solver's temporaries on top of that spilled
// 300-480 registers per lane (k_scaling_slow), and the image held nine copies of a 10 k-instruction loop
#define LTP_MR_SOLVER __device__ __attribute__((noinline))
#else
// host build (tests only): the helpers of ltp_math.hpp this file uses
namespace ltp {
static inline double dabs(double x) { return __builtin_fabs(x); }
static inline bool dfinite(double x) { return dabs(x) < __builtin_huge_val(); }
static inline bool disnan(double x) { return x != x; }
static inline double dsqrt(double x) { return __builtin_sqrt(x); }
static inline double dmax(double a, double b) { return a < b ? b : a; }
static inline double dmin(double a, double b) { return b < a ? b : a; }
}  // namespace ltp
#define LTP_MR_FN static inline
#define LTP_MR_SOLVER static __attribute__((noinline))
#endif

namespace ltp {
namespace mr {

constexpr int kMaxN = 6;
constexpr double kDblMinM = 2.2250738585072014e-308;
constexpr double kDblEpsM = 2.220446049250313e-16;
constexpr double kDblMaxM = 1.7976931348623157e+308;

// f(constant I) for I = B .. E-1 (UP) or E-1 .. B (down): the loop index is a constant expression inside f
template <int I> struct Idx { static constexpr int value = I; };
template <int B, int E, class F>
LTP_MR_FN void static_for(F&& f)
{
    if constexpr (B < E) {
        f(Idx<B>{});
        static_for<B + 1, E>(f);
    }
}
template <int B, int E, class F>
LTP_MR_FN void static_for_down(F&& f)
{
    if constexpr (B < E) {
        f(Idx<E - 1>{});
        static_for_down<B, E - 1>(f);
    }
}
#define LTP_MR_I(ic) (decltype(ic)::value)
// every loop body is a lambda; left to the inliner's cost model some stayed calls until after the last scalar-replacement pass, and the
// arrays they capture by reference (a row of the matrix, the eigenvalue arrays) stayed in scratch memory: loads and stores on the
// critical path of every sweep
#define LTP_MR_INL __attribute__((always_inline))

// c ? a : b on VALUES. (With two variables as operands the conditional operator yields one of the two OBJECTS — the compiler selects an
// address and loads through it, which takes the address of a matrix element and keeps it in scratch memory.)
LTP_MR_FN double pick(bool c, double a, double b) { return c ? a : b; }

LTP_MR_FN double fsign(double a, double b) { return __builtin_signbit(b) ? -dabs(a) : dabs(a); }   // Fortran SIGN(a, b)

LTP_MR_FN double lapy2(double x, double y)
{
    const double xa = dabs(x), ya = dabs(y);
    const double w = dmax(xa, ya), z = dmin(xa, ya);
    if (disnan(x)) return x;
    if (disnan(y)) return y;
    if (z == 0.0 || w > kDblMaxM) return w;
    return w * dsqrt(1.0 + (z / w) * (z / w));
}

// one term of DNRM2's scaled sum of squares: "if (scale < a) { ssq = 1 + ssq (scale / a)^2; scale = a } else ssq += (a / scale)^2" with
// the quotient of either branch formed by ONE division (lanes of a wave take different branches: two divisions would both be run)
LTP_MR_FN void nrm2_term(double x, double& scale, double& ssq)
{
    const double a = dabs(x);
    if (a != 0.0) {
        const bool up = scale < a;
        const double t = pick(up, scale, a) / pick(up, a, scale);
        if (up) { ssq = 1.0 + ssq * t * t; scale = a; }
        else ssq += t * t;
    }
}
// the first term, on (scale, ssq) = (0, 1): a finite or infinite a > 0 gives ssq = 1 + 1 (0 / a)^2 = 1 exactly and scale = a; a NaN
// fails "scale < a" and makes ssq NaN. No division.
LTP_MR_FN void nrm2_first_term(double x, double& scale, double& ssq)
{
    const double a = dabs(x);
    if (a != 0.0) {
        if (a == a) scale = a;
        else ssq = a;
    }
}
// DNRM2 of v[1 .. NR-1]. One element: scale sqrt(ssq) with (scale, ssq) = (|v1|, 1), (0, 1) or (0, NaN), i.e. |v1| in every case.
template <int NR>
LTP_MR_FN double tail_nrm2(double v1, double v2)
{
    if constexpr (NR == 2) {
        return dabs(v1);
    } else {
        double scale = 0.0, ssq = 1.0;
        nrm2_first_term(v1, scale, ssq);
        nrm2_term(v2, scale, ssq);
        return scale * dsqrt(ssq);
    }
}

// DLARFG on (v0; v1 [, v2]), NR = 2 or 3 (the vector as scalars: an array handed down by reference stayed in scratch memory, three
// loads and stores per reflector on the critical path of every sweep)
template <int NR>
LTP_MR_FN void larfg(double& v0, double& v1, double& v2, double& tau)
{
    static_assert(NR == 2 || NR == 3, "reflectors of a double-shift sweep");
    const double safmin = kDblMinM / (kDblEpsM * 0.5);
    double alpha = v0;
    double xnorm = tail_nrm2<NR>(v1, v2);
    if (xnorm == 0.0) { tau = 0.0; return; }
    double beta = -fsign(lapy2(alpha, xnorm), alpha);
    int knt = 0;
    if (dabs(beta) < safmin) {
        const double rsafmn = 1.0 / safmin;
        do {
            ++knt;
            v1 *= rsafmn;
            if constexpr (NR == 3) v2 *= rsafmn;
            beta *= rsafmn;
            alpha *= rsafmn;
        } while (dabs(beta) < safmin && knt < 20);
        xnorm = tail_nrm2<NR>(v1, v2);
        beta = -fsign(lapy2(alpha, xnorm), alpha);
    }
    tau = (beta - alpha) / beta;
    const double s = 1.0 / (alpha - beta);
    v1 *= s;
    if constexpr (NR == 3) v2 *= s;
    for (int x = 0; x < knt; ++x) beta *= safmin;
    v0 = beta;
}

// DLANV2: eigenvalues of [a b; c d]; (rt1r, rt1i) first
LTP_MR_FN void lanv2(double a, double b, double c, double d, double& rt1r, double& rt1i, double& rt2r, double& rt2i)
{
    const double multpl = 4.0, eps = kDblEpsM;
    if (c == 0.0) {
    } else if (b == 0.0) {
        const double temp = d;
        d = a; a = temp; b = -c; c = 0.0;
    } else if ((a - d) == 0.0 && fsign(1.0, b) != fsign(1.0, c)) {
    } else {
        double temp = a - d;
        double p = 0.5 * temp;
        const double bcmax = dmax(dabs(b), dabs(c));
        const double bcmis = dmin(dabs(b), dabs(c)) * fsign(1.0, b) * fsign(1.0, c);
        const double scale = dmax(dabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= multpl * eps) {
            z = p + fsign(dsqrt(scale) * dsqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            b = b - c;
            c = 0.0;
        } else {
            const double sigma = b + c;
            p = 0.5 * temp;
            const double tau = lapy2(sigma, temp);
            const double cs = dsqrt(0.5 * (1.0 + dabs(sigma) / tau));
            const double sn = -(p / (tau * cs)) * fsign(1.0, sigma);
            const double aa = a * cs + b * sn, bb = -a * sn + b * cs;
            const double cc = c * cs + d * sn, dd = -c * sn + d * cs;
            a = aa * cs + cc * sn;
            b = bb * cs + dd * sn;
            c = -aa * sn + cc * cs;
            d = -bb * sn + dd * cs;
            temp = 0.5 * (a + d);
            a = temp;
            d = temp;
            if (c != 0.0) {
                if (b != 0.0) {
                    if (fsign(1.0, b) == fsign(1.0, c)) {
                        const double sab = dsqrt(dabs(b)), sac = dsqrt(dabs(c));
                        p = fsign(sab * sac, c);
                        a = temp + p;
                        d = temp - p;
                        b = b - c;
                        c = 0.0;
                    }
                } else {
                    b = -c;
                    c = 0.0;
                }
            }
        }
    }
    rt1r = a;
    rt2r = d;
    if (c == 0.0) {
        rt1i = 0.0;
        rt2i = 0.0;
    } else {
        rt1i = dsqrt(dabs(b)) * dsqrt(dabs(c));
        rt2i = -rt1i;
    }
}

// DGEBAL('B') without the permutation search, in place. Rows and columns beyond the matrix order are zero: LAPACK's loop over
// i < n is the loop over i < N here (a zero row or column is skipped by DGEBAL's own "c == 0 or r == 0" rule), its sums over j < n
// the sums over j < N (DNRM2 skips zeros; max with |0| changes nothing).
template <int N>
LTP_MR_FN void balance(double (&T)[N][N])
{
    const double sclfac = 2.0, factor = 0.95;
    const double sfmin1 = kDblMinM / kDblEpsM, sfmax1 = 1.0 / sfmin1;
    const double sfmin2 = sfmin1 * sclfac, sfmax2 = 1.0 / sfmin2;
    double scalev[N];
    static_for<0, N>([&](auto ic) LTP_MR_INL { scalev[LTP_MR_I(ic)] = 1.0; });
    bool noconv = true, nan = false;
    for (int guard = 0; noconv && guard < 1000; ++guard) {
        noconv = false;
        static_for<0, N>([&](auto ic) LTP_MR_INL {
            constexpr int i = LTP_MR_I(ic);
            if (nan) return;
            double cs = 0.0, cq = 1.0, rs = 0.0, rq = 1.0, ca = 0.0, ra = 0.0;
            static_for<0, N>([&](auto jc) LTP_MR_INL { nrm2_term(T[LTP_MR_I(jc)][i], cs, cq); });
            static_for<0, N>([&](auto jc) LTP_MR_INL { nrm2_term(T[i][LTP_MR_I(jc)], rs, rq); });
            double c = cs * dsqrt(cq), r = rs * dsqrt(rq);
            static_for<0, N>([&](auto jc) LTP_MR_INL {
                ca = dmax(ca, dabs(T[LTP_MR_I(jc)][i]));
                ra = dmax(ra, dabs(T[i][LTP_MR_I(jc)]));
            });
            if (c == 0.0 || r == 0.0) return;
            double g = r / sclfac, f = 1.0;
            const double s = c + r;
            while (!(c >= g || dmax(f, dmax(c, ca)) >= sfmax2 || dmin(r, dmin(g, ra)) <= sfmin2)) {
                if (disnan(c + f + ca + r + g + ra)) { nan = true; break; }
                f *= sclfac; c *= sclfac; ca *= sclfac;
                r /= sclfac; g /= sclfac; ra /= sclfac;
            }
            if (nan) return;
            g = c / sclfac;
            while (!(g < r || dmax(r, ra) >= sfmax2 || dmin(dmin(f, c), dmin(g, ca)) <= sfmin2)) {
                f /= sclfac; c /= sclfac; g /= sclfac; ca /= sclfac;
                r *= sclfac; ra *= sclfac;
            }
            if (c + r >= factor * s) return;
            if (f < 1.0 && scalev[i] < 1.0 && f * scalev[i] <= sfmin1) return;
            if (f > 1.0 && scalev[i] > 1.0 && scalev[i] >= sfmax1 / f) return;
            g = 1.0 / f;
            scalev[i] *= f;
            noconv = true;
            static_for<0, N>([&](auto jc) LTP_MR_INL { T[i][LTP_MR_I(jc)] *= g; });
            static_for<0, N>([&](auto jc) LTP_MR_INL { T[LTP_MR_I(jc)][i] *= f; });
        });
        if (nan) return;
    }
}

// DLAHQR, eigenvalues only (WANTT = WANTZ = false), on the leading n x n part of T (upper Hessenberg, zero elsewhere). Returns 0, or
// i + 1 if the iteration did not converge. LAPACK's two nested loops (over the active block's bottom row i, over the iterations of
// one block) run as ONE loop here: each pass looks for a negligible sub-diagonal entry in rows l+1 .. i and then either takes the
// one or two eigenvalues that split off at the bottom or performs one sweep. wr / wi: N entries, those not reached stay untouched.
template <int N>
LTP_MR_FN int lahqr(double (&T)[N][N], int n, double (&wr)[N], double (&wi)[N])
{
    const double dat1 = 3.0 / 4.0, dat2 = -0.4375;
    const int kexsh = 10;
    const double safmin = kDblMinM, ulp = kDblEpsM;
    const double smlnum = safmin * ((double)n / ulp);
    const int itmax = 30 * 10;                                         // 30 * max(n, 10), n <= 6
    // (DLAHQR first clears the entries below the sub-diagonal: they are zero by construction and DGEBAL only scales)
    int i = n - 1, l = 0, kdefl = 0, its = 0;
    while (i >= 0) {
        // look for a single small subdiagonal element: LAPACK's k = i, i-1, ..., l+1, the first k that passes a test; else l
        {
            int kk = l;
            bool stop = false;
            static_for_down<1, N>([&](auto rc) LTP_MR_INL {
                constexpr int r = LTP_MR_I(rc);
                if (stop || r > i || r <= l) return;
                const double sub = dabs(T[r][r - 1]);
                bool brk = sub <= smlnum;
                if (!brk) {
                    double tst = dabs(T[r - 1][r - 1]) + dabs(T[r][r]);
                    if (tst == 0.0) {
                        if constexpr (r - 2 >= 0) tst += dabs(T[r - 1][r - 2]);
                        if constexpr (r + 1 <= N - 1) tst += dabs(T[r + 1][r]);        // zero from row n on
                    }
                    if (sub <= ulp * tst) {
                        const double sup = dabs(T[r - 1][r]);
                        const double ab = dmax(sub, sup);
                        const double ba = dmin(sub, sup);
                        const double dd = dabs(T[r - 1][r - 1] - T[r][r]);
                        const double aa = dmax(dabs(T[r][r]), dd);
                        const double bb = dmin(dabs(T[r][r]), dd);
                        const double ss = aa + ab;
                        if (ba * (ab / ss) <= dmax(smlnum, ulp * (bb * (aa / ss)))) brk = true;
                    }
                }
                if (brk) { kk = r; stop = true; }
            });
            l = kk;
        }
        // (an element at a position only known at run time is read and written as a SELECT of values over the constant positions, never
        // under a branch: the optimizer merges "if (u == i) x = T[u][u]" over u into one load through a pointer chosen at run time, and an
        // element whose address is taken that way stays in scratch memory)
        static_for<1, N>([&](auto rc) LTP_MR_INL {
            constexpr int r = LTP_MR_I(rc);
            T[r][r - 1] = pick(r == l, 0.0, T[r][r - 1]);
        });
        // the bottom 2 x 2 of the active block, wherever it is
        double hmm = 0.0, hmi = 0.0, him = 0.0, hii = 0.0;
        static_for<0, N>([&](auto uc) LTP_MR_INL {
            constexpr int u = LTP_MR_I(uc);
            const bool at = u == i;
            hii = pick(at, T[u][u], hii);
            if constexpr (u >= 1) {
                hmm = pick(at, T[u - 1][u - 1], hmm);
                hmi = pick(at, T[u - 1][u], hmi);
                him = pick(at, T[u][u - 1], him);
            }
        });
        if (l >= i - 1) {
            // one or two eigenvalues have split off
            double r1r = 0.0, r1i = 0.0, r2r = 0.0, r2i = 0.0;
            const bool pair = l != i;
            if (pair) lanv2(hmm, hmi, him, hii, r1r, r1i, r2r, r2i);
            else r1r = hii;
            static_for<0, N>([&](auto uc) LTP_MR_INL {
                constexpr int u = LTP_MR_I(uc);
                const bool first = pair ? u == i - 1 : u == i, second = pair && u == i;
                wr[u] = pick(first, r1r, pick(second, r2r, wr[u]));
                wi[u] = pick(first, r1i, pick(second, r2i, wi[u]));
            });
            kdefl = 0;
            its = 0;
            i = l - 1;
            l = 0;
            continue;
        }
        ++kdefl;
        double h11, h21, h12, h22, s;
        if (kdefl % kexsh == 0) {
            // exceptional shifts (every 10th / 20th iteration without a deflation)
            double d0 = 0.0, s0 = 0.0, s1 = 0.0;
            if (kdefl % (2 * kexsh) == 0) {
                static_for<2, N>([&](auto uc) LTP_MR_INL {
                    constexpr int u = LTP_MR_I(uc);
                    const bool at = u == i;
                    d0 = pick(at, T[u][u], d0); s0 = pick(at, T[u][u - 1], s0); s1 = pick(at, T[u - 1][u - 2], s1);
                });
            } else {
                static_for<0, N - 2>([&](auto uc) LTP_MR_INL {
                    constexpr int u = LTP_MR_I(uc);
                    const bool at = u == l;
                    d0 = pick(at, T[u][u], d0); s0 = pick(at, T[u + 1][u], s0); s1 = pick(at, T[u + 2][u + 1], s1);
                });
            }
            s = dabs(s0) + dabs(s1);
            h11 = dat1 * s + d0;
            h12 = dat2 * s;
            h21 = s;
            h22 = h11;
        } else {
            h11 = hmm;
            h21 = him;
            h12 = hmi;
            h22 = hii;
        }
        s = dabs(h11) + dabs(h12) + dabs(h21) + dabs(h22);
        double rt1r = 0.0, rt1i = 0.0, rt2r = 0.0, rt2i = 0.0;
        if (s != 0.0) {
            h11 /= s; h21 /= s; h12 /= s; h22 /= s;
            const double tr = (h11 + h22) / 2.0;
            const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
            const double rtdisc = dsqrt(dabs(det));
            if (det >= 0.0) {
                rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i;
            } else {
                rt1r = tr + rtdisc;
                rt2r = tr - rtdisc;
                if (dabs(rt1r - h22) <= dabs(rt2r - h22)) { rt1r = rt1r * s; rt2r = rt1r; }
                else { rt2r = rt2r * s; rt1r = rt2r; }
            }
        }
        // look for two consecutive small subdiagonal elements: m = i-2, ..., l
        double v0 = 0.0, v1 = 0.0, v2 = 0.0;
        int m = l;
        {
            bool stop = false;
            static_for_down<0, N - 2>([&](auto mc) LTP_MR_INL {
                constexpr int mm = LTP_MR_I(mc);
                if (stop || mm > i - 2 || mm < l) return;
                double h21s = dabs(T[mm + 1][mm]);
                double sc = dabs(T[mm][mm] - rt2r) + dabs(rt2i) + h21s;
                h21s = T[mm + 1][mm] / sc;
                v0 = h21s * T[mm][mm + 1] + (T[mm][mm] - rt1r) * ((T[mm][mm] - rt2r) / sc) - rt1i * (rt2i / sc);
                v1 = h21s * (T[mm][mm] + T[mm + 1][mm + 1] - rt1r - rt2r);
                v2 = h21s * T[mm + 2][mm + 1];
                sc = dabs(v0) + dabs(v1) + dabs(v2);
                v0 /= sc; v1 /= sc; v2 /= sc;
                m = mm;
                if (mm == l) { stop = true; return; }
                if constexpr (mm >= 1) {
                    const double h00 = dabs(T[mm - 1][mm - 1]), hh11 = dabs(T[mm][mm]), hh22 = dabs(T[mm + 1][mm + 1]);
                    if (dabs(T[mm][mm - 1]) * (dabs(v1) + dabs(v2)) <= ulp * dabs(v0) * (h00 + hh11 + hh22)) stop = true;
                }
            });
        }
        // double-shift QR sweep on rows / columns l .. i (eigenvalues only: i1 = l, i2 = i): k = m .. i-1
        static_for<0, N - 1>([&](auto kc) LTP_MR_INL {
            constexpr int k = LTP_MR_I(kc);
            if (k < m || k > i - 1) return;
            const bool three = k < i - 1;                              // nr = min(3, i - k + 1)
            double t1;
            if constexpr (k >= 1) {
                if (k > m) {
                    v0 = T[k][k - 1];
                    v1 = T[k + 1][k - 1];
                    if constexpr (k + 2 <= N - 1) { if (three) v2 = T[k + 2][k - 1]; }
                }
            }
            if constexpr (k + 2 <= N - 1) {
                if (three) larfg<3>(v0, v1, v2, t1);
                else larfg<2>(v0, v1, v2, t1);
            } else {
                larfg<2>(v0, v1, v2, t1);                                     // the last row pair: never three rows
            }
            if (k > m) {
                if constexpr (k >= 1) {
                    T[k][k - 1] = v0;
                    T[k + 1][k - 1] = 0.0;
                    if constexpr (k + 2 <= N - 1) { if (three) T[k + 2][k - 1] = 0.0; }
                }
            } else if (m > l) {
                if constexpr (k >= 1) T[k][k - 1] = T[k][k - 1] * (1.0 - t1);
            }
            const double w2 = v1, t2 = t1 * w2;
            if (three) {
                if constexpr (k + 2 <= N - 1) {
                    const double w3 = v2, t3 = t1 * w3;
                    static_for<k, N>([&](auto jc) LTP_MR_INL {
                        constexpr int j = LTP_MR_I(jc);
                        if (j > i) return;
                        const double sum = T[k][j] + w2 * T[k + 1][j] + w3 * T[k + 2][j];
                        T[k][j] -= sum * t1;
                        T[k + 1][j] -= sum * t2;
                        T[k + 2][j] -= sum * t3;
                    });
                    static_for<0, (k + 3 < N - 1 ? k + 3 : N - 1) + 1>([&](auto jc) LTP_MR_INL {
                        constexpr int j = LTP_MR_I(jc);
                        if (j < l || j > i) return;                    // j = l .. min(k + 3, i)
                        const double sum = T[j][k] + w2 * T[j][k + 1] + w3 * T[j][k + 2];
                        T[j][k] -= sum * t1;
                        T[j][k + 1] -= sum * t2;
                        T[j][k + 2] -= sum * t3;
                    });
                }
            } else {
                // nr == 2: k == i - 1
                static_for<k, N>([&](auto jc) LTP_MR_INL {
                    constexpr int j = LTP_MR_I(jc);
                    if (j > i) return;
                    const double sum = T[k][j] + w2 * T[k + 1][j];
                    T[k][j] -= sum * t1;
                    T[k + 1][j] -= sum * t2;
                });
                static_for<0, k + 2>([&](auto jc) LTP_MR_INL {
                    constexpr int j = LTP_MR_I(jc);
                    if (j < l) return;                                 // j = l .. i
                    const double sum = T[j][k] + w2 * T[j][k + 1];
                    T[j][k] -= sum * t1;
                    T[j][k + 1] -= sum * t2;
                });
            }
        });
        ++its;
        if (its > itmax) return i + 1;
    }
    return 0;
}

// roots(c) for c[0 .. N], N <= 6, highest coefficient first. re / im: N entries in MATLAB's output order (zero roots from stripped
// trailing zero coefficients first); nroots = N minus the stripped leading zeros. 0 = ok, 1 = no convergence, 2 = NaN / Inf
// coefficient (MATLAB: error).
template <int N>
LTP_MR_SOLVER int roots_n(const double (&c)[N + 1], double (&re)[kMaxN], double (&im)[kMaxN], int& nroots)
{
    static_assert(N >= 1 && N <= kMaxN, "degree");
    const double nan = __builtin_nan("");
    static_for<0, N>([&](auto ic) LTP_MR_INL { re[LTP_MR_I(ic)] = nan; im[LTP_MR_I(ic)] = nan; });
    nroots = 0;
    bool finite = true;
    static_for<0, N + 1>([&](auto ic) LTP_MR_INL { finite = finite && dfinite(c[LTP_MR_I(ic)]); });
    if (!finite) return 2;
    int first = 0;
    {
        bool lead = true;
        static_for<0, N + 1>([&](auto ic) LTP_MR_INL {
            if (lead && c[LTP_MR_I(ic)] == 0.0) ++first;
            else lead = false;
        });
    }
    if (first > N) return 0;
    int last = N;
    {
        bool trail = true;
        static_for_down<1, N + 1>([&](auto ic) LTP_MR_INL {
            if (trail && LTP_MR_I(ic) > first && c[LTP_MR_I(ic)] == 0.0) --last;
            else trail = false;
        });
    }
    const int n = last - first;
    nroots = N - first;
    // d[j] = c[first + j]
    double d[N + 1];
    static_for<0, N + 1>([&](auto jc) LTP_MR_INL { d[LTP_MR_I(jc)] = 0.0; });
    static_for<0, N + 1>([&](auto fc) LTP_MR_INL {
        constexpr int f = LTP_MR_I(fc);
        if (first == f) static_for<0, N + 1 - f>([&](auto jc) LTP_MR_INL { d[LTP_MR_I(jc)] = c[f + LTP_MR_I(jc)]; });
    });
    double T[N][N];
    static_for<0, N>([&](auto ic) LTP_MR_INL { static_for<0, N>([&](auto jc) LTP_MR_INL { T[LTP_MR_I(ic)][LTP_MR_I(jc)] = 0.0; }); });
    static_for<1, N>([&](auto ic) LTP_MR_INL {
        constexpr int i = LTP_MR_I(ic);
        if (i < n) T[i][i - 1] = 1.0;
    });
    static_for<0, N>([&](auto jc) LTP_MR_INL {
        constexpr int j = LTP_MR_I(jc);
        if (j < n) {
            T[0][j] = -d[1 + j] / d[0];
            finite = finite && dfinite(T[0][j]);
        }
    });
    if (!finite) return 2;
    balance<N>(T);
    double wr[N], wi[N];
    static_for<0, N>([&](auto ic) LTP_MR_INL { wr[LTP_MR_I(ic)] = nan; wi[LTP_MR_I(ic)] = nan; });
    const int info = lahqr<N>(T, n, wr, wi);
    const int nz = nroots - n;                                         // zero roots come first: r = [zeros(nnz, 1); eig(A)]
    static_for<0, N>([&](auto kc) LTP_MR_INL {
        constexpr int k = LTP_MR_I(kc);
        double rk = nan, ik = nan;                                     // (entries behind the last root stay NaN)
        static_for<0, k + 1>([&](auto uc) LTP_MR_INL {
            constexpr int u = LTP_MR_I(uc);
            const bool at = k - nz == u && u < n;
            rk = pick(at, wr[u], rk);
            ik = pick(at, wi[u], ik);
        });
        re[k] = pick(k < nz, 0.0, rk);
        im[k] = pick(k < nz, 0.0, ik);
    });
    return info ? 1 : 0;
}

// the same with the degree as a value (tests)
LTP_MR_FN int roots(const double* c, int deg, double (&re)[kMaxN], double (&im)[kMaxN], int& nroots)
{
    const double nan = __builtin_nan("");
    nroots = 0;
    if (deg < 0 || deg > kMaxN) {
        for (int i = 0; i < deg && i < kMaxN; ++i) { re[i] = nan; im[i] = nan; }
        return 2;
    }
    if (deg == 0) return dfinite(c[0]) ? 0 : 2;
    int st = 2;
    static_for<1, kMaxN + 1>([&](auto dc) LTP_MR_INL {
        constexpr int D = LTP_MR_I(dc);
        if (deg == D) {
            double cc[D + 1];
            static_for<0, D + 1>([&](auto ic) LTP_MR_INL { cc[LTP_MR_I(ic)] = c[LTP_MR_I(ic)]; });
            st = roots_n<D>(cc, re, im, nroots);
        }
    });
    return st;
}

}  // namespace mr
}  // namespace ltp
