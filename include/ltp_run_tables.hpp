// ltp_run_tables.hpp — PUBLIC device-side interface to the planner's run tables: the on-device consumer hook
// (SURVEY.md §8(f).2, §7 hard part 5). HIP / gfx950, header-only, no dependency on the rest of the library.
//
// Why. The reference's getTrajectory (src/long_term_planner.cc:706-841) returns four dense arrays per joint; 1 M 7-DoF plans
// at 1 ms are 386 GB, more than a GPU holds. A consumer that reduces the trajectory (an envelope, a peak velocity, a time above
// a threshold, a closest approach ...) does not need the rows: it needs, per joint, the <= 20 RUNS into which the sampler cuts
// the trajectory — inside a run every stored sample is a closed-form polynomial of the sample's position in the run — and the
// function that evaluates them. ltp_build_tables_batch (include/ltp_hip.h) leaves exactly that in a caller buffer, 912 bytes
// per (plan, joint); this header is the format and the arithmetic, the SAME single-source functions the library's own samplers
// and consumers are built from (csrc/ltp_sampler*.hip, ltp_consumers.hip include this file): whatever a consumer computes with
// them has the bits of the rows ltp_sample_batch would have stored.
//
// What a run is. Run r of a joint covers trajectory samples [start(r), start(r + 1)) (0-based, start(runs) = traj_len). With
// m = i - start(r) + 1 the 1-based position of sample i in its run,
//     q(i) = fma(fma(fma(c[3], m, c[2]), m, c[1]), m, c[0])     v(i) = fma(fma(c[6], m, c[5]), m, c[4])
//     a(i) = fma(c[8], m, c[7])                                 j(i) = c[9]
// (run_eval). The ten coefficients follow from five stored words — the state (a, v, q) before the run, its jerk and its mode
// bits — plus the joint's snap velocity v_drive * dir (cc:823), by run_coef(); they encode the reference's recurrence
// cc:810-831 including its three snap rules (cc:815-829), which only change coefficients.
//
// Two ways to consume:
//   (1) one lane per (plan, joint), straight from global memory: RunTableView + for_each_run / for_each_sample. Simple; right for
//       reductions whose cost per joint is small or whose lanes do similar work.
//   (2) block-cooperative, tables staged in LDS: fetch_run_tables + install_run_tables fill JointTable[<= 8] in LDS for one plan
//       (256 threads); every lane then reads any joint's runs (RunCursor). This is how the library's envelope consumer
//       (k_envelope) is written, and what to use when many lanes share one joint (windows, sample ranges).
// tests/cpp/example_consumer.hip is a complete user-side consumer (compiled with plain hipcc against this header only).
//
// Global layout (what ltp_build_tables_batch writes). Lane index i = local_plan * dof + joint. Lanes are stored in tiles of 64;
// inside a tile the 16-byte word PAIR (2k, 2k + 1) of lane l sits at pair k, lane l: word w of lane i is the 8-byte element
// table_word_index(i, w). Per lane kPackedWords = 114 words:
//   word 0          runs (low 32 bits) | traj_len (high 32 bits); runs == 0: the plan has no trajectory (failed / rejected)
//   words 1 .. 11   int start[22]: first sample of run r; start[runs] = traj_len; (start[21] is used by the library's capped-row
//                   sampler for a row offset and is not part of the contract)
//   word 12         vsnap = v_drive * dir (cc:823); word 13 unused
//   words 14 + 5 r .. 18 + 5 r   run r: a, v, q before the run, its jerk (doubles), its mode bits (kMode*, low 32 bits)
#pragma once
#include <hip/hip_runtime.h>

namespace ltp {

#ifndef LTP_DEV
#define LTP_DEV __device__ __forceinline__
#endif
#ifndef LTP_HD
#define LTP_HD __host__ __device__ inline
#endif

// Which source the arithmetic follows where the C++ reference and the MATLAB original diverge (include/ltp_hip.h,
// ltp_set_semantics). run_coef<kSemMatlab> understands a superset of the C++ mode bits and gives the same bits for them, so a
// consumer of tables may always use it (the tables carry the modes they were built with).
constexpr int kSemCpp = 0, kSemMatlab = 1;

constexpr int kMaxSegments = 20;       // runs of constant jerk and mode per joint (1 + 19 cut points)
constexpr int kRunCoefs = 10;          // q0..q3, v0..v2, a0, a1, J (monomial basis in m)
constexpr int kModeTail = 1;           // i > s6: a = 0, v = 0 (cc:815-829)
constexpr int kModeVSnap = 2;          // phase 4 interior: v = v_drive*dir (cc:822-823)
constexpr int kModeKeepA = 4;          // MATLAB semantics, tail of every joint but the last: a keeps following the jerk sums (LTPlanner.m:607)

// Inside one run, with m = 1-based position in the run, the reference's recurrence forms
//   a(m) = a_s + m Ts J,   v(m) = v_s + Ts (m a_s + Ts J m(m+1)/2),   q(m) = q_s + Ts (m v_s + Ts (a_s m(m+1)/2 + Ts J m(m+1)(m+2)/6)),
// i.e. polynomials of degree 1, 2 and 3 in m. They are stored in the monomial basis and evaluated by Horner's rule with fused
// multiply-adds — 6 arithmetic instructions per sample. The stage kernels keep the reference's unfused operation order (branch
// decisions hang on it); here any rounding order is ~1e-12 from the sequential sums (bar 1e-9).
struct RunCoef {
    double c[kRunCoefs];
    int mode;                    // the kMode* bits the coefficients were made with (the table pass stores them)
};

// coefficients of a run that starts after state (a_s, v_s, q_s)
template <int SEM = kSemCpp>
LTP_DEV RunCoef run_coef(int mode, double J, double a_s, double v_s, double q_s, double vsnap, double Ts)
{
    // The products and sums below are separate roundings in the library (it is built with -ffp-contract=off); pinned here so that
    // a consumer compiled with hipcc's default (-ffp-contract=fast) gets the same coefficients, i.e. the library's bits.
#pragma clang fp contract(off)
    RunCoef r;
#pragma unroll
    for (int x = 0; x < kRunCoefs; ++x) r.c[x] = 0.0;
    const double tj = Ts * J;
    r.mode = mode;
    r.c[9] = J;
    if (!(mode & kModeTail) || (SEM == kSemMatlab && (mode & kModeKeepA))) { r.c[7] = a_s; r.c[8] = tj; }
    r.c[0] = q_s;
    if (mode & kModeVSnap) {
        r.c[4] = vsnap;
        r.c[1] = Ts * vsnap;
    } else if (!(mode & kModeTail)) {
        // binomial-sum form -> monomial basis: m(m+1)/2 = (m^2 + m)/2, m(m+1)(m+2)/6 = (m^3 + 3 m^2 + 2 m)/6
        const double v1 = Ts * a_s, v2 = 0.5 * (Ts * tj);
        const double q1 = Ts * v_s, q2 = 0.5 * (Ts * (Ts * a_s)), q3 = (Ts * (Ts * tj)) * (1.0 / 6.0);
        r.c[4] = v_s; r.c[5] = v1 + v2; r.c[6] = v2;
        r.c[1] = q1 + (q2 + 2.0 * q3); r.c[2] = q2 + 3.0 * q3; r.c[3] = q3;
    }
    return r;
}

// run_coef for callers whose lanes hold runs of DIFFERENT kinds (a lane-per-joint walk, ltp_runs.hpp): the same coefficients, to the
// bit, from selects of the inputs instead of branches around the coefficient blocks — a wave whose lanes sit in all three kinds of
// run would execute every branch in turn. A constant-velocity run is an ordinary run whose velocity is vsnap and whose acceleration
// and jerk contribute +0.0 to v and q (x + 0.0 == x; v_drive is never zero), a tail run one whose state is all zero. Callers
// whose waves hold ONE run at a time (the streaming loops: the mode is wave-uniform, the untaken branches are skipped) use run_coef.
template <int SEM = kSemCpp>
LTP_DEV RunCoef run_coef_sel(int mode, double J, double a_s, double v_s, double q_s, double vsnap, double Ts)
{
#pragma clang fp contract(off)
    RunCoef r;
    const bool tail = (mode & kModeTail) != 0, vsn = (mode & kModeVSnap) != 0;
    const bool keep_a = !tail || (SEM == kSemMatlab && (mode & kModeKeepA));
    const bool plain = !tail && !vsn;
    const double tj = Ts * J;
    r.mode = mode;
    r.c[9] = J;
    r.c[7] = keep_a ? a_s : 0.0;
    r.c[8] = keep_a ? tj : 0.0;
    r.c[0] = q_s;
    const double a_v = plain ? a_s : 0.0, tj_v = plain ? tj : 0.0;
    const double v_b = vsn ? vsnap : (plain ? v_s : 0.0);
    // binomial-sum form -> monomial basis: m(m+1)/2 = (m^2 + m)/2, m(m+1)(m+2)/6 = (m^3 + 3 m^2 + 2 m)/6
    const double v1 = Ts * a_v, v2 = 0.5 * (Ts * tj_v);
    const double q1 = Ts * v_b, q2 = 0.5 * (Ts * (Ts * a_v)), q3 = (Ts * (Ts * tj_v)) * (1.0 / 6.0);
    r.c[4] = v_b; r.c[5] = v1 + v2; r.c[6] = v2;
    r.c[1] = q1 + (q2 + 2.0 * q3); r.c[2] = q2 + 3.0 * q3; r.c[3] = q3;
    return r;
}

// the four outputs at position m of a run; the streaming loops and the state propagation both use exactly this
LTP_DEV void run_eval(const double (&c)[kRunCoefs], int m, double& q, double& v, double& a, double& j)
{
    const double md = (double)m;
    q = __builtin_fma(__builtin_fma(__builtin_fma(c[3], md, c[2]), md, c[1]), md, c[0]);
    v = __builtin_fma(__builtin_fma(c[6], md, c[5]), md, c[4]);
    a = __builtin_fma(c[8], md, c[7]);
    j = c[9];
}
LTP_DEV double run_eval_q(const double* c, int m)
{
    const double md = (double)m;
    return __builtin_fma(__builtin_fma(__builtin_fma(c[3], md, c[2]), md, c[1]), md, c[0]);   // the q line of run_eval
}

// ---------------------------------------------------------------------------------------
// Global-memory form (packed): see the layout at the top of this file.
// ---------------------------------------------------------------------------------------
constexpr int kPackedHeaderWords = 14;
constexpr int kPackedRunWords = 5;
constexpr int kPackedWords = kPackedHeaderWords + kMaxSegments * kPackedRunWords;           // 114
static_assert(kPackedWords % 2 == 0 && kPackedHeaderWords % 2 == 0, "word pairs");
static_assert(kPackedWords / 2 <= 64, "one 16-byte-per-lane load instruction brings a whole packed table");

LTP_HD unsigned long long table_word_index(unsigned long long lane, int word)
{
    return (lane >> 6) * (unsigned long long)(kPackedWords * 64) + ((unsigned long long)(word >> 1) * 64ull + (lane & 63ull)) * 2ull + (unsigned long long)(word & 1);
}
// bytes of the tables of `lanes` (plan, joint) lanes (whole tiles of 64 lanes)
LTP_HD unsigned long long run_table_bytes(long long lanes)
{
    return (unsigned long long)((lanes + 63) / 64) * (unsigned long long)kPackedWords * 64ull * 8ull;
}

// (1) One lane, straight from global memory.
struct RunTableView {
    const unsigned long long* tables;     // what ltp_build_tables_batch filled
    unsigned long long lane;              // local_plan * dof + joint
    LTP_DEV unsigned long long word(int w) const { return tables[table_word_index(lane, w)]; }
    LTP_DEV int runs() const { return (int)(unsigned)word(0); }                    // 0: no trajectory
    LTP_DEV int traj_len() const { return (int)(unsigned)(word(0) >> 32); }
    LTP_DEV int start(int r) const { return (int)(unsigned)(word(1 + (r >> 1)) >> (32 * (r & 1))); }   // r <= runs(): start(runs()) = traj_len
    LTP_DEV double vsnap() const { return __builtin_bit_cast(double, word(12)); }
    LTP_DEV RunCoef coef(int r, double vs, double Ts) const
    {
        const int w0 = kPackedHeaderWords + kPackedRunWords * r;
        return run_coef<kSemMatlab>((int)(unsigned)word(w0 + 4), __builtin_bit_cast(double, word(w0 + 3)), __builtin_bit_cast(double, word(w0)),
                                    __builtin_bit_cast(double, word(w0 + 1)), __builtin_bit_cast(double, word(w0 + 2)), vs, Ts);
    }
};

// visit(b, e, rc) for every run [b, e) of the joint with its coefficients, in order; return true from visit to stop.
// Ts: the sample time the batch was planned with (ltp_get_sample_time).
template <class Visit>
LTP_DEV void for_each_run(const RunTableView& t, double Ts, Visit&& visit)
{
    const int n = t.runs();
    if (n <= 0) return;
    const double vs = t.vsnap();
    int b = t.start(0);
    for (int r = 0; r < n; ++r) {
        const int e = t.start(r + 1);
        if (visit(b, e, t.coef(r, vs, Ts))) return;
        b = e;
    }
}

// visit(i, q, v, a, j) for the trajectory samples first <= i < last (clamped to the trajectory), in order: exactly the values
// ltp_sample_batch would have stored at sample i of this joint.
template <class Visit>
LTP_DEV void for_each_sample(const RunTableView& t, double Ts, int first, int last, Visit&& visit)
{
    for_each_run(t, Ts, [&](int b, int e, const RunCoef& rc) {
        const int lo = b > first ? b : first, hi = e < last ? e : last;
        for (int i = lo; i < hi; ++i) {
            double q, v, a, j;
            run_eval(rc.c, i - b + 1, q, v, a, j);
            visit(i, q, v, a, j);
        }
        return e >= last;
    });
}

// ---------------------------------------------------------------------------------------
// (2) Block-cooperative form: the tables of one plan x up to kRunTableJoints joints expanded in LDS.
// ---------------------------------------------------------------------------------------
constexpr int kRunTableThreads = 256;  // block size fetch_run_tables / install_run_tables are written for
constexpr int kRunTableJoints = 8;     // joints (consecutive lanes of one plan) staged per call

// The run table of one joint in LDS: kTableWords 8-byte words.
struct JointTable {
    int nseg;                               // word 0 (low half): runs
    int len;                                // traj_len of the plan (tables from the table pass; a fused build leaves it unset)
    int start[kMaxSegments + 2];            // words 1..11: first sample of run k; start[nseg] = traj_len (capped tables: the first
                                            // run that is not stored); start[kMaxSegments + 1]: library-internal (row offset)
    double c[kMaxSegments][kRunCoefs];      // words 12..211: the coefficients run_eval takes
};
constexpr int kTableWords = 1 + (kMaxSegments + 2) / 2 + kMaxSegments * kRunCoefs;
static_assert(sizeof(JointTable) == kTableWords * 8, "JointTable must be kTableWords 8-byte words");
// In LDS the packed words land at the END of the JointTable they expand into (install_run_tables reads everything it expands
// before anything is overwritten).
constexpr int kPackedAt = kTableWords - kPackedWords;                                       // first JointTable word of the packed form
static_assert(kPackedAt % 2 == 0, "16-byte aligned in LDS");
constexpr int kTableLoads = (kPackedWords + 31) / 32;

// What one thread of a 256-thread block holds of the packed tables of `nj` consecutive lanes starting at `lane0` (the joints
// j0 .. j0 + nj - 1 of one plan): word (threadIdx.x >> 3) + 32 r of joint slot threadIdx.x & 7. Issues the loads and returns;
// nothing here waits for them, so a caller can overlap them with other work before install_run_tables.
struct PackedTableRegs {
    unsigned long long w[kTableLoads];
};
LTP_DEV PackedTableRegs fetch_run_tables(const unsigned long long* __restrict__ tables, unsigned long long lane0, int nj)
{
    PackedTableRegs r;
#pragma unroll
    for (int x = 0; x < kTableLoads; ++x) r.w[x] = 0ull;
    const int jt = threadIdx.x & 7, wb = threadIdx.x >> 3;
    if (jt < nj) {
        const unsigned long long lane = lane0 + (unsigned long long)jt;
#pragma unroll
        for (int x = 0; x < kTableLoads; ++x) {
            const int w = wb + 32 * x;
            if (w < kPackedWords) r.w[x] = tables[table_word_index(lane, w)];
        }
    }
    return r;
}

// Places the fetched packed tables in jt[0 .. nj) (LDS) and expands them in place. Every thread of the 256-thread block calls
// this; it ends with the block barrier after which any thread may read the tables.
LTP_DEV void install_run_tables(JointTable* jt, int nj, const unsigned long long (&w)[kTableLoads], double Ts)
{
    static_assert(kRunTableJoints * kMaxSegments <= kRunTableThreads && kRunTableJoints * (kPackedHeaderWords - 2) <= kRunTableThreads, "one task per thread");
    {
        const int js = threadIdx.x & 7, wb = threadIdx.x >> 3;
        if (js < nj) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(&jt[js]) + kPackedAt;
#pragma unroll
            for (int x = 0; x < kTableLoads; ++x) {
                const int word = wb + 32 * x;
                if (word < kPackedWords) dst[word] = w[x];
            }
        }
    }
    __syncthreads();
    // every thread reads what it expands (the coefficients of run r overwrite packed words of later runs), then all write
    const int t = threadIdx.x;
    const int jx = t / kMaxSegments, r = t - jx * kMaxSegments;          // run task
    const int hx = t / (kPackedHeaderWords - 2), hw = t - hx * (kPackedHeaderWords - 2);   // header word task
    unsigned long long hdr = 0ull;
    if (hx < nj) hdr = reinterpret_cast<const unsigned long long*>(&jt[hx])[kPackedAt + hw];
    bool live = false;
    RunCoef rc;
    if (jx < nj) {
        const unsigned long long* pk = reinterpret_cast<const unsigned long long*>(&jt[jx]) + kPackedAt;
        const int nseg = (int)(unsigned)pk[0];
        if (r < nseg) {
            live = true;
            const double* st = reinterpret_cast<const double*>(pk + kPackedHeaderWords + r * kPackedRunWords);
            rc = run_coef<kSemMatlab>((int)(unsigned)pk[kPackedHeaderWords + r * kPackedRunWords + 4], st[3], st[0], st[1], st[2],
                                      reinterpret_cast<const double*>(pk)[12], Ts);   // (a superset of the C++ modes: same bits)
        }
    }
    __syncthreads();
    if (hx < nj) reinterpret_cast<unsigned long long*>(&jt[hx])[hw] = hdr;
    if (live) {
#pragma unroll
        for (int x = 0; x < kRunCoefs; ++x) jt[jx].c[r][x] = rc.c[x];
    }
    __syncthreads();
}

// A lane's position in one joint's runs while it walks samples in increasing order: sample i lies in run `run`, which covers
// [cur, nxt) (nxt = INT_MAX for the last run); position in the run m = i - cur + 1. advance(i) returns true when the run changed
// (re-read the coefficients jt.c[run] then).
struct RunCursor {
    int run, cur, nxt;
    LTP_DEV explicit RunCursor(const JointTable& jt) : run(0), cur(0), nxt(jt.nseg > 1 ? jt.start[1] : 0x7fffffff) {}
    LTP_DEV bool advance(const JointTable& jt, int i)
    {
        if (nxt > i) return false;
        do {
            ++run;
            cur = nxt;
            nxt = run + 1 < jt.nseg ? jt.start[run + 1] : 0x7fffffff;
        } while (nxt <= i);
        return true;
    }
};

}  // namespace ltp
