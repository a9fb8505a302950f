// matlab_roots_test.cc — the register-resident restatement of MATLAB's roots() that the device runs in MATLAB-semantics mode
// (longtermplanner_amd/csrc/ltp_roots_matlab.hpp) against the oracle's loop-form twin (oracle/matlab_roots.inc, ltpm_roots in
// libltp_oracle.so), on the host. Plain g++, no GPU, no HIP: the header is host/device-portable; tests/test_gpu_matlab.py makes the
// same comparison on the device.
//   usage: matlab_roots_test <thousands of polynomials per class and degree> <seed>     exit code 0 = every output bit-identical
// (status, nroots and all re / im entries, NaN == NaN). -ffp-contract=off as the library is built.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../longtermplanner_amd/csrc/ltp_roots_matlab.hpp"

extern "C" int ltpm_roots(const double* c, int deg, double* re, double* im, int* nroots);
extern "C" int ltpm_debug_iters(int* max_kdefl);      // sweeps of the twin's last solve, and the longest stretch without a deflation

static uint64_t splitmix(uint64_t& s)
{
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
static double u01(uint64_t& s) { return (double)(splitmix(s) >> 11) * 0x1p-53; }
static double sym(uint64_t& s) { return 2.0 * u01(s) - 1.0; }
static uint64_t bits(double x) { uint64_t u; memcpy(&u, &x, 8); return u; }
static bool same(double a, double b) { return bits(a) == bits(b) || (a != a && b != b); }

static long long g_bad = 0, g_n = 0, g_noconv = 0, g_stripped = 0, g_exc10 = 0, g_exc20 = 0, g_sweeps = 0;

static void check(const double* c, int deg)
{
    double wre[6], wim[6], gre[6], gim[6];
    for (int i = 0; i < 6; ++i) wre[i] = wim[i] = gre[i] = gim[i] = -777.0;
    int wn = -1, gn = -1;
    const int wst = ltpm_roots(c, deg, wre, wim, &wn);
    const int gst = ltp::mr::roots(c, deg, gre, gim, gn);
    ++g_n;
    {
        int maxk = 0;
        g_sweeps += ltpm_debug_iters(&maxk);
        if (maxk >= 10) ++g_exc10;
        if (maxk >= 20) ++g_exc20;
    }
    if (wst == 1) ++g_noconv;
    if (wn != deg || (deg > 0 && c[deg] == 0.0)) ++g_stripped;
    bool ok = wst == gst && wn == gn;
    for (int i = 0; i < deg && ok; ++i) ok = same(wre[i], gre[i]) && same(wim[i], gim[i]);
    if (!ok) {
        if (g_bad < 10) {
            fprintf(stderr, "MISMATCH deg %d: status %d / %d, nroots %d / %d\n  c =", deg, wst, gst, wn, gn);
            for (int i = 0; i <= deg; ++i) fprintf(stderr, " %a", c[i]);
            fprintf(stderr, "\n");
            for (int i = 0; i < deg; ++i) fprintf(stderr, "  [%d] twin %a %+ai   header %a %+ai\n", i, wre[i], wim[i], gre[i], gim[i]);
        }
        ++g_bad;
    }
}

// coefficients of prod (x - r_k) with real roots and conjugate pairs
static void from_roots(const double* rr, const double* ri, int deg, double* c)
{
    double p[8] = {1.0};
    int d = 0;
    for (int k = 0; k < deg;) {
        if (ri[k] != 0.0 && k + 1 < deg) {
            const double b = -2.0 * rr[k], cc = rr[k] * rr[k] + ri[k] * ri[k];      // x^2 + b x + cc
            double q[8] = {0};
            for (int i = 0; i <= d; ++i) { q[i] += p[i]; q[i + 1] += b * p[i]; q[i + 2] += cc * p[i]; }
            d += 2;
            memcpy(p, q, sizeof p);
            k += 2;
        } else {
            double q[8] = {0};
            for (int i = 0; i <= d; ++i) { q[i] += p[i]; q[i + 1] += -rr[k] * p[i]; }
            d += 1;
            memcpy(p, q, sizeof p);
            k += 1;
        }
    }
    for (int i = 0; i <= deg; ++i) c[i] = p[i];
}

int main(int argc, char** argv)
{
    const long long per = (argc > 1 ? atoll(argv[1]) : 200) * 1000ll;
    uint64_t s = argc > 2 ? strtoull(argv[2], nullptr, 0) : 4242;
    for (int deg = 0; deg <= 6; ++deg) {
        double c[7];
        for (long long it = 0; it < per; ++it) {
            // class 1: coefficients uniform in [-1, 1]
            for (int i = 0; i <= deg; ++i) c[i] = sym(s);
            check(c, deg);
            // class 2: log-uniform magnitudes over 1e-12 .. 1e12 (the planner's polynomials mix times, jerks and their powers)
            for (int i = 0; i <= deg; ++i) c[i] = (splitmix(s) & 1 ? -1.0 : 1.0) * exp((u01(s) * 2.0 - 1.0) * 27.6);
            check(c, deg);
            // class 3: from chosen roots — clusters, exact multiples and conjugate pairs next to the real axis (slow convergence:
            // the exceptional shifts of iterations 10 and 20 are taken here)
            {
                double rr[6], ri[6];
                const double centre = sym(s) * 3.0, spread = exp(-u01(s) * 30.0);
                for (int k = 0; k < deg; ++k) {
                    const unsigned pick = (unsigned)(splitmix(s) % 4);
                    rr[k] = pick == 0 ? centre : pick == 1 ? centre + spread * sym(s) : sym(s) * 10.0;
                    ri[k] = 0.0;
                }
                for (int k = 0; k + 1 < deg; k += 2) {
                    if (splitmix(s) % 3 == 0) { ri[k] = spread * u01(s); ri[k + 1] = -ri[k]; rr[k + 1] = rr[k]; }
                }
                from_roots(rr, ri, deg, c);
                const double lead = exp(sym(s) * 10.0);
                for (int i = 0; i <= deg; ++i) c[i] *= lead;
                check(c, deg);
            }
            // class 4: leading and trailing zero coefficients (stripped by roots.m), zeros inside
            if (it % 4 == 0) {
                for (int i = 0; i <= deg; ++i) c[i] = sym(s);
                const int lead0 = (int)(splitmix(s) % (unsigned)(deg + 2)), trail0 = (int)(splitmix(s) % (unsigned)(deg + 2));
                for (int i = 0; i < lead0 && i <= deg; ++i) c[i] = (splitmix(s) & 7) ? 0.0 : -0.0;
                for (int i = 0; i < trail0 && i <= deg; ++i) c[deg - i] = 0.0;
                if (deg >= 2 && (splitmix(s) & 1)) c[1 + splitmix(s) % (unsigned)(deg - 1)] = 0.0;
                check(c, deg);
            }
            // class 5: extreme magnitudes (balancing by many powers of two, overflowing quotients, subnormals), NaN / Inf
            if (it % 8 == 0) {
                for (int i = 0; i <= deg; ++i) c[i] = sym(s) * exp2((double)((int)(splitmix(s) % 2000) - 1000));
                check(c, deg);
                for (int i = 0; i <= deg; ++i) c[i] = sym(s) * exp2((double)((int)(splitmix(s) % 120) - 1074 + 60));
                check(c, deg);
                if (it % 64 == 0) {
                    for (int i = 0; i <= deg; ++i) c[i] = sym(s);
                    c[splitmix(s) % (unsigned)(deg + 1)] = (splitmix(s) & 1) ? NAN : INFINITY;
                    check(c, deg);
                }
            }
        }
    }
    printf("{\"polynomials\": %lld, \"mismatches\": %lld, \"sweeps\": %lld, \"with_an_exceptional_shift\": %lld, \"with_both_exceptional_shifts\": %lld, \"not_converged\": %lld, \"with_stripped_zeros\": %lld}\n",
           g_n, g_bad, g_sweeps, g_exc10, g_exc20, g_noconv, g_stripped);
    return g_bad == 0 ? 0 : 1;
}
