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
// in the test suite (tests/test_matlab_twin.py) and on the device against that twin (tests/test_gpu_matlab.py); against MATLAB's own
// LAPACK build it is unpinned.
//
// One lane solves one polynomial; unlike the register-resident solver of ltp_roots.hpp the matrix is indexed dynamically. Round 4: it
// lives in DYNAMIC LDS (element (i, j) of thread t at word (6 i + j) * T + t, T = threads per block: conflict-free whatever the lanes
// index), not in scratch memory — same operations, same bits. Every launch of a kernel that can reach roots() in this mode passes
// matrix_lds_bytes(threads per block) of dynamic shared memory (ltp_stage_kernels.hip, ltp_aux_kernels.hip).
#pragma once
#include "ltp_math.hpp"

namespace ltp {
namespace mr {

constexpr int kMaxN = 6;
constexpr double kDblMinM = 2.2250738585072014e-308;
constexpr double kDblEpsM = 2.220446049250313e-16;
constexpr double kDblMaxM = 1.7976931348623157e+308;

struct LdsMat {                           // this thread's matrix inside the block's LDS image
    double* p;
    int stride;                           // threads per block
    LTP_DEV double& operator()(int i, int j) const { return p[(i * kMaxN + j) * stride]; }
};
// dynamic shared memory a launch of `threads` threads per block must provide for the matrices
__host__ __device__ inline unsigned matrix_lds_bytes(int threads) { return (unsigned)(kMaxN * kMaxN * threads) * (unsigned)sizeof(double); }

LTP_DEV double fsign(double a, double b) { return __builtin_signbit(b) ? -dabs(a) : dabs(a); }   // Fortran SIGN(a, b)

LTP_DEV double lapy2(double x, double y)
{
    const double xa = dabs(x), ya = dabs(y);
    const double w = dmax(xa, ya), z = dmin(xa, ya);
    if (disnan(x)) return x;
    if (disnan(y)) return y;
    if (z == 0.0 || w > kDblMaxM) return w;
    return w * dsqrt(1.0 + (z / w) * (z / w));
}

// DNRM2 (scaled sum of squares) of n values with a stride
LTP_DEV double nrm2(int n, const double* x, int inc)
{
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; ++i) {
        const double a = dabs(x[i * inc]);
        if (a != 0.0) {
            if (scale < a) { ssq = 1.0 + ssq * (scale / a) * (scale / a); scale = a; }
            else ssq += (a / scale) * (a / scale);
        }
    }
    return scale * dsqrt(ssq);
}

// DNRM2 of column `col` (ROW == false) or row `row` (ROW == true) of the matrix: nrm2() above with the matrix accessor
template <bool ROW, class M>
LTP_DEV double mat_nrm2(int n, M& H, int k)
{
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < n; ++i) {
        const double a = dabs(ROW ? H(k, i) : H(i, k));
        if (a != 0.0) {
            if (scale < a) { ssq = 1.0 + ssq * (scale / a) * (scale / a); scale = a; }
            else ssq += (a / scale) * (a / scale);
        }
    }
    return scale * dsqrt(ssq);
}

// DLARFG, nr = 2 or 3
LTP_DEV void larfg(int nr, double* v, double& tau)
{
    const double safmin = kDblMinM / (kDblEpsM * 0.5);
    double alpha = v[0];
    if (nr <= 1) { tau = 0.0; return; }
    double xnorm = nrm2(nr - 1, v + 1, 1);
    if (xnorm == 0.0) { tau = 0.0; return; }
    double beta = -fsign(lapy2(alpha, xnorm), alpha);
    int knt = 0;
    if (dabs(beta) < safmin) {
        const double rsafmn = 1.0 / safmin;
        do {
            ++knt;
            for (int i = 1; i < nr; ++i) v[i] *= rsafmn;
            beta *= rsafmn;
            alpha *= rsafmn;
        } while (dabs(beta) < safmin && knt < 20);
        xnorm = nrm2(nr - 1, v + 1, 1);
        beta = -fsign(lapy2(alpha, xnorm), alpha);
    }
    tau = (beta - alpha) / beta;
    const double s = 1.0 / (alpha - beta);
    for (int i = 1; i < nr; ++i) v[i] *= s;
    for (int i = 0; i < knt; ++i) beta *= safmin;
    v[0] = beta;
}

// DLANV2: eigenvalues of [a b; c d]; (rt1r, rt1i) first
LTP_DEV void lanv2(double a, double b, double c, double d, double& rt1r, double& rt1i, double& rt2r, double& rt2i)
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

// DGEBAL('B') without the permutation search, in place
template <class M>
LTP_DEV void balance(int n, M& H)
{
    const double sclfac = 2.0, factor = 0.95;
    const double sfmin1 = kDblMinM / kDblEpsM, sfmax1 = 1.0 / sfmin1;
    const double sfmin2 = sfmin1 * sclfac, sfmax2 = 1.0 / sfmin2;
    double scalev[kMaxN];
    for (int i = 0; i < n; ++i) scalev[i] = 1.0;
    bool noconv = true;
    for (int guard = 0; noconv && guard < 1000; ++guard) {
        noconv = false;
        for (int i = 0; i < n; ++i) {
            double c = mat_nrm2<false>(n, H, i), r = mat_nrm2<true>(n, H, i), ca = 0.0, ra = 0.0;
            for (int j = 0; j < n; ++j) { ca = dmax(ca, dabs(H(j, i))); ra = dmax(ra, dabs(H(i, j))); }
            if (c == 0.0 || r == 0.0) continue;
            double g = r / sclfac, f = 1.0;
            const double s = c + r;
            bool nan = false;
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
            if (c + r >= factor * s) continue;
            if (f < 1.0 && scalev[i] < 1.0 && f * scalev[i] <= sfmin1) continue;
            if (f > 1.0 && scalev[i] > 1.0 && scalev[i] >= sfmax1 / f) continue;
            g = 1.0 / f;
            scalev[i] *= f;
            noconv = true;
            for (int j = 0; j < n; ++j) H(i, j) *= g;
            for (int j = 0; j < n; ++j) H(j, i) *= f;
        }
    }
}

// DLAHQR, eigenvalues only. Returns 0, or i + 1 if the iteration did not converge.
template <class M>
LTP_DEV int lahqr(int n, M& H, double* wr, double* wi)
{
    const double dat1 = 3.0 / 4.0, dat2 = -0.4375;
    const int kexsh = 10;
    const double safmin = kDblMinM, ulp = kDblEpsM;
    const double smlnum = safmin * ((double)n / ulp);
    const int itmax = 30 * (n > 10 ? n : 10);
    if (n == 0) return 0;
    if (n == 1) { wr[0] = H(0, 0); wi[0] = 0.0; return 0; }
    for (int j = 0; j <= n - 4; ++j) { H(j + 2, j) = 0.0; H(j + 3, j) = 0.0; }
    if (n >= 3) H(n - 1, n - 3) = 0.0;
    int i = n - 1, kdefl = 0;
    while (i >= 0) {
        bool converged = false;
        int l = 0;
        for (int its = 0; its <= itmax; ++its) {
            int k;
            for (k = i; k > l; --k) {
                if (dabs(H(k, k - 1)) <= smlnum) break;
                double tst = dabs(H(k - 1, k - 1)) + dabs(H(k, k));
                if (tst == 0.0) {
                    if (k - 2 >= 0) tst += dabs(H(k - 1, k - 2));
                    if (k + 1 <= n - 1) tst += dabs(H(k + 1, k));
                }
                if (dabs(H(k, k - 1)) <= ulp * tst) {
                    const double ab = dmax(dabs(H(k, k - 1)), dabs(H(k - 1, k)));
                    const double ba = dmin(dabs(H(k, k - 1)), dabs(H(k - 1, k)));
                    const double aa = dmax(dabs(H(k, k)), dabs(H(k - 1, k - 1) - H(k, k)));
                    const double bb = dmin(dabs(H(k, k)), dabs(H(k - 1, k - 1) - H(k, k)));
                    const double ss = aa + ab;
                    if (ba * (ab / ss) <= dmax(smlnum, ulp * (bb * (aa / ss)))) break;
                }
            }
            l = k;
            if (l > 0) H(l, l - 1) = 0.0;
            if (l >= i - 1) { converged = true; break; }
            ++kdefl;
            double h11, h21, h12, h22, s;
            if (kdefl % (2 * kexsh) == 0) {
                s = dabs(H(i, i - 1)) + dabs(H(i - 1, i - 2));
                h11 = dat1 * s + H(i, i);
                h12 = dat2 * s;
                h21 = s;
                h22 = h11;
            } else if (kdefl % kexsh == 0) {
                s = dabs(H(l + 1, l)) + dabs(H(l + 2, l + 1));
                h11 = dat1 * s + H(l, l);
                h12 = dat2 * s;
                h21 = s;
                h22 = h11;
            } else {
                h11 = H(i - 1, i - 1);
                h21 = H(i, i - 1);
                h12 = H(i - 1, i);
                h22 = H(i, i);
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
            double v[3] = {0.0, 0.0, 0.0};
            int m;
            for (m = i - 2; m >= l; --m) {
                double h21s = dabs(H(m + 1, m));
                s = dabs(H(m, m) - rt2r) + dabs(rt2i) + h21s;
                h21s = H(m + 1, m) / s;
                v[0] = h21s * H(m, m + 1) + (H(m, m) - rt1r) * ((H(m, m) - rt2r) / s) - rt1i * (rt2i / s);
                v[1] = h21s * (H(m, m) + H(m + 1, m + 1) - rt1r - rt2r);
                v[2] = h21s * H(m + 2, m + 1);
                s = dabs(v[0]) + dabs(v[1]) + dabs(v[2]);
                v[0] /= s; v[1] /= s; v[2] /= s;
                if (m == l) break;
                const double h00 = dabs(H(m - 1, m - 1)), hh11 = dabs(H(m, m)), hh22 = dabs(H(m + 1, m + 1));
                if (dabs(H(m, m - 1)) * (dabs(v[1]) + dabs(v[2])) <= ulp * dabs(v[0]) * (h00 + hh11 + hh22)) break;
            }
            for (k = m; k <= i - 1; ++k) {
                const int nr = (3 < i - k + 1) ? 3 : i - k + 1;
                double t1;
                if (k > m) for (int e = 0; e < nr; ++e) v[e] = H(k + e, k - 1);
                larfg(nr, v, t1);
                if (k > m) {
                    H(k, k - 1) = v[0];
                    H(k + 1, k - 1) = 0.0;
                    if (k < i - 1) H(k + 2, k - 1) = 0.0;
                } else if (m > l) {
                    H(k, k - 1) = H(k, k - 1) * (1.0 - t1);
                }
                const double v2 = v[1], t2 = t1 * v2;
                if (nr == 3) {
                    const double v3 = v[2], t3 = t1 * v3;
                    const int jend = (k + 3 < i) ? k + 3 : i;
                    for (int j = k; j <= i; ++j) {
                        const double sum = H(k, j) + v2 * H(k + 1, j) + v3 * H(k + 2, j);
                        H(k, j) -= sum * t1;
                        H(k + 1, j) -= sum * t2;
                        H(k + 2, j) -= sum * t3;
                    }
                    for (int j = l; j <= jend; ++j) {
                        const double sum = H(j, k) + v2 * H(j, k + 1) + v3 * H(j, k + 2);
                        H(j, k) -= sum * t1;
                        H(j, k + 1) -= sum * t2;
                        H(j, k + 2) -= sum * t3;
                    }
                } else if (nr == 2) {
                    for (int j = k; j <= i; ++j) {
                        const double sum = H(k, j) + v2 * H(k + 1, j);
                        H(k, j) -= sum * t1;
                        H(k + 1, j) -= sum * t2;
                    }
                    for (int j = l; j <= i; ++j) {
                        const double sum = H(j, k) + v2 * H(j, k + 1);
                        H(j, k) -= sum * t1;
                        H(j, k + 1) -= sum * t2;
                    }
                }
            }
        }
        if (!converged) return i + 1;
        if (l == i) {
            wr[i] = H(i, i);
            wi[i] = 0.0;
        } else {
            lanv2(H(i - 1, i - 1), H(i - 1, i), H(i, i - 1), H(i, i), wr[i - 1], wi[i - 1], wr[i], wi[i]);
        }
        kdefl = 0;
        i = l - 1;
    }
    return 0;
}

// roots(c) for c[0..deg], deg <= 6, highest coefficient first. re / im: deg entries in MATLAB's output order (zero roots
// from stripped trailing zero coefficients first); nroots = deg minus the stripped leading zeros. 0 = ok, 1 = no
// convergence, 2 = NaN / Inf coefficient (MATLAB: error).
template <class M>
LTP_DEV int roots_with(M& H, const double* c, int deg, double* re, double* im, int& nroots)
{
    const double nan = __builtin_nan("");
    for (int i = 0; i < deg; ++i) { re[i] = nan; im[i] = nan; }
    nroots = 0;
    if (deg < 0 || deg > kMaxN) return 2;
    for (int i = 0; i <= deg; ++i) if (!dfinite(c[i])) return 2;
    int first = 0, last = deg;
    while (first <= deg && c[first] == 0.0) ++first;
    if (first > deg) return 0;
    while (last > first && c[last] == 0.0) --last;
    const int n = last - first;
    nroots = deg - first;
    for (int i = 0; i < kMaxN; ++i) for (int j = 0; j < kMaxN; ++j) H(i, j) = 0.0;
    for (int i = 1; i < n; ++i) H(i, i - 1) = 1.0;
    for (int j = 0; j < n; ++j) H(0, j) = -c[first + 1 + j] / c[first];
    for (int j = 0; j < n; ++j) if (!dfinite(H(0, j))) return 2;
    balance(n, H);
    const int nz = nroots - n;
    for (int i = 0; i < nz; ++i) { re[i] = 0.0; im[i] = 0.0; }
    return lahqr(n, H, re + nz, im + nz) ? 1 : 0;
}

LTP_DEV int roots(const double* c, int deg, double* re, double* im, int& nroots)
{
    extern __shared__ double ltp_mr_matrices[];               // matrix_lds_bytes(threads per block), passed by every launch that gets here
    const int threads = (int)(blockDim.x * blockDim.y * blockDim.z);
    // (round-4 advisor) the convention is checked, not assumed: the launch's LDS allocation (dispatch packet, group_segment_size) must hold
    // the kernel's static LDS plus this block's matrices — a launch that forgot the dynamic size traps instead of overwriting LDS data
    const unsigned group_bytes = ((const unsigned __attribute__((address_space(4)))*)__builtin_amdgcn_dispatch_ptr())[7];   // byte 28
    if (__builtin_amdgcn_groupstaticsize() + matrix_lds_bytes(threads) > group_bytes) __builtin_trap();
    const int tid = ((int)threadIdx.z * (int)blockDim.y + (int)threadIdx.y) * (int)blockDim.x + (int)threadIdx.x;
    LdsMat H{ltp_mr_matrices + tid, threads};
    return roots_with(H, c, deg, re, im, nroots);
}

}  // namespace mr
}  // namespace ltp
