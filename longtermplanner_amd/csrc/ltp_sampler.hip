// ltp_sampler.hip — getTrajectory (cc:706-841) as the HBM-bound sampler k_sample, and the kernels that share its run
// tables or run walk (k_envelope, k_replan_states, k_state_at), gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile). No fast-math:
// the inf/NaN flow of the reference is part of the contract.
#include "ltp_device.hpp"

namespace ltp {

// ---------------------------------------------------------------------------------------
// The sampler: reference getTrajectory (cc:706-841).
//
// The reference integrates a piecewise-constant jerk sample by sample. Here the jerk array of
// one joint (seven range fills cc:759-766, then up to eight "+=" fractional corrections
// cc:768-807) is cut at every index where it, or one of the three snap rules (cc:815-829),
// can change: at most 20 runs of constant jerk and constant mode. The block builds the list of
// runs cooperatively, one lane per joint walks the runs once and leaves the state before each
// run's first sample in LDS, and after that every sample is independent: within a run that
// starts after state (a_s, v_s, q_s),
//     a[m] = a_s + m*Ts*J
//     v[m] = v_s + Ts*(m*a_s + Ts*J*m(m+1)/2)
//     q[m] = q_s + Ts*(m*v_s + Ts*(a_s*m(m+1)/2 + Ts*J*m(m+1)(m+2)/6))
// are exactly the sums the recurrence forms (up to the order of rounding, ~1e-13), so all 256
// lanes stream q/v/a/j rows to HBM as 16-byte stores, 1 KiB contiguous per wave instruction.
// ---------------------------------------------------------------------------------------
constexpr int kModeTail = 1;    // i > s6: a = 0, v = 0 (cc:815-829)
constexpr int kModeVSnap = 2;   // phase 4 interior: v = v_drive*dir (cc:822-823)
constexpr int kModeKeepA = 4;   // MATLAB semantics, tail of every joint but the last: a keeps following the jerk sums (LTPlanner.m:607)

// Inside one run, with m = 1-based position in the run, the reference's recurrence forms
//   a(m) = a_s + m Ts J,   v(m) = v_s + Ts (m a_s + Ts J m(m+1)/2),   q(m) = q_s + Ts (m v_s + Ts (a_s m(m+1)/2 + Ts J m(m+1)(m+2)/6)),
// i.e. polynomials of degree 1, 2 and 3 in m. They are stored in the monomial basis and evaluated by Horner's rule with fused
// multiply-adds,
//   q(m) = fma(fma(fma(q3, m, q2), m, q1), m, q0),   v(m) = fma(fma(v2, m, v1), m, v0),   a(m) = fma(a1, m, a0),   j(m) = J
// — 6 arithmetic instructions per sample. That is what bounds the rows that carry more samples per byte (float32 rows: round 2
// measured 75 % VALU-busy at 0.74 of the HBM peak with 19 instructions per sample) and the envelope consumer. The stage kernels
// keep the reference's unfused operation order (branch decisions hang on it); here any rounding order is ~1e-12 from the
// sequential sums (bar 1e-9). The three snap rules of cc:815-829 only change coefficients, so evaluating a sample has no branches.
constexpr int kRunCoefs = 10;   // q0..q3, v0..v2, a0, a1, J (monomial basis in m)
struct RunCoef {
    double c[kRunCoefs];
    int mode;                    // the kMode* bits the coefficients were made with (the table pass stores them)
};

struct SegScratch {              // scratch of the cooperative table build, dead once the coefficients are written
    int s[kSampleJointGroup][8];            // sampled switch indices (cc:751-757)
    double fr[kSampleJointGroup][8];        // fractions lost to sampling (cc:747)
    double frts[kSampleJointGroup][8];      // fr / Ts
    double misc[kSampleJointGroup][8];      // dir*j_max, v_drive*dir, q_0, v_0, a_0, mod
    double Jp[kSampleJointGroup][8];        // jerk of the seven phases (cc:735-744)
    double corr[kSampleJointGroup][10];     // the nine possible "+=" correction terms (cc:771-807)
    int cand[kSampleJointGroup][kMaxSegments];
    double runJ[kSampleJointGroup][kMaxSegments];
    int runMode[kSampleJointGroup][kMaxSegments];
    double state[kSampleJointGroup][kMaxSegments][3];
};
// The run table of one joint: kTableWords 8-byte words. This is the layout in LDS; the table pass (k_build_tables) keeps a
// packed form of it in global memory for the sampler variants that do not build tables themselves (below).
struct JointTable {
    int nseg;                               // word 0 (low half)
    int len;                                // table pass only: traj_len of the plan (the fused build leaves it unset)
    int start[kMaxSegments + 2];            // words 1..11: first sample of run k; start[nseg] = traj_len (or the first run
                                            // that is not stored); table pass only: start[kMaxSegments + 1] = the plan's row
                                            // offset inside the sampled range, in units of kRowAlign elements
    double c[kMaxSegments][kRunCoefs];      // words 12..211
};
constexpr int kTableWords = 1 + (kMaxSegments + 2) / 2 + kMaxSegments * kRunCoefs;
static_assert(sizeof(JointTable) == kTableWords * 8, "JointTable must be kTableWords 8-byte words");
// What the table pass keeps in global memory is the PACKED form of a JointTable: the header as it is, then per run the five
// words run_coef() makes the ten coefficients from — half the bytes to write and to read back; the consumer expands them with
// the same run_coef() (same operations, same bits):
//   words 0..11   nseg | len, start[]                       (JointTable words 0..11)
//   word  12      vsnap = v_drive * dir (cc:823); word 13 unused
//   words 14 + 5r .. 18 + 5r   a, v, q before run r, its jerk, its mode bits
// In LDS the packed words land at the END of the JointTable they expand into (unpack order: see expand_packed_tables).
constexpr int kPackedHeaderWords = 14;
constexpr int kPackedRunWords = 5;
constexpr int kPackedWords = kPackedHeaderWords + kMaxSegments * kPackedRunWords;           // 114
constexpr int kPackedAt = kTableWords - kPackedWords;                                       // first JointTable word of the packed form
static_assert(kPackedWords % 2 == 0 && kPackedAt % 2 == 0 && kPackedHeaderWords % 2 == 0, "word pairs, 16-byte aligned in LDS");
static_assert(kPackedWords / 2 <= 64, "one LDS-direct load instruction brings a whole packed table");
struct SegTable {
    JointTable jt[kSampleJointGroup];
    union {
        SegScratch w;
        // the sampler reuses the space for the finished 16-byte slots that contain run boundary k: [q, v, a, j]
        double2_t bnd[kSampleJointGroup][kMaxSegments][4];
    };
};

// candidate cut points: slot 0 is index 0, slot c >= 1 is s[kCutBase[c]] + kCutDelta[c]; every index where the jerk
// array or a snap rule (cc:815-829) can change is among them
constexpr int kCutSlots = 20;
__device__ const signed char kCutBase[kCutSlots] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 6};
__device__ const signed char kCutDelta[kCutSlots] = {0, 0, 1, 2, 0, 1, 0, 1, 2, -1, 0, 1, 0, 1, 2, 0, 1, 0, 1, 2};

// value of the reference's j_traj[joint][i] after the seven range fills (cc:759-766, last writer wins) and the up
// to eight "+=" fractional corrections (cc:768-807), applied in the reference's order and association (cc:781 and
// cc:798 add two / three terms to the element one after the other). s = sampled switch indices,
// Jp = jerk of the seven phases, corr = the correction terms, all in LDS.
template <int SEM = kSemCpp>
LTP_DEV double jerk_at(const int* s, const double* Jp, const double* corr, int i)
{
    const int s0 = s[0], s1 = s[1], s2 = s[2], s3 = s[3], s4 = s[4], s5 = s[5], s6 = s[6];
    double val = 0.0;
    if (s0 > 0 && i < s0) val = Jp[0];
    if (s1 - s0 > 0 && i >= s0 && i < s1) val = Jp[1];
    if (s2 - s1 > 0 && i >= s1 && i < s2) val = Jp[2];
    if (s3 - s2 > 0 && i >= s2 && i < s3) val = Jp[3];
    if (s4 - s3 > 0 && i >= s3 && i < s4) val = Jp[4];
    if (s5 - s4 > 0 && i >= s4 && i < s5) val = Jp[5];
    if (s6 - s5 > 0 && i >= s5 && i < s6) val = Jp[6];
    // LTPlanner.m:558-597 addresses the same elements in a 1-BASED array: every correction lands one sample earlier than in
    // the C++, which kept the index expressions for its 0-based arrays
    constexpr int o = SEM == kSemMatlab ? 1 : 0;
    if (s2 >= s1) {
        if (i == s0 + 1 - o) val = val + corr[0];
        if (s1 > 0 && i == s1 - o) val = val + corr[1];
        if (i == s2 + 1 - o) val = val + corr[2];
    } else {
        if (s1 > 0 && i == s1 - o) val = (val + corr[0]) + corr[3];             // cc:781: j + A + B, left to right
    }
    if (s3 > 0 && i == s3 - o) val = val + corr[4];
    if (s2 - s0 > 0) {
        if (i == s4 + 1 - o) val = val + corr[5];
    } else {
        if (s4 > 0 && i == s4 - o) val = ((val + corr[5]) + corr[0]) + corr[3]; // cc:798: j + A + B + C, left to right
    }
    if (s5 > 0 && i == s5 - o) val = val + corr[7];
    if (i == s6 + 1 - o) val = val + corr[8];
    return val;
}

// coefficients of a run that starts after state (a_s, v_s, q_s)
template <int SEM = kSemCpp>
LTP_DEV RunCoef run_coef(int mode, double J, double a_s, double v_s, double q_s, double vsnap, double Ts)
{
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

// the four outputs at position m of a run; the streaming loop and the state propagation both use exactly this
LTP_DEV void run_eval(const double (&c)[kRunCoefs], int m, double& q, double& v, double& a, double& j)
{
    const double md = (double)m;
    q = __builtin_fma(__builtin_fma(__builtin_fma(c[3], md, c[2]), md, c[1]), md, c[0]);
    v = __builtin_fma(__builtin_fma(c[6], md, c[5]), md, c[4]);
    a = __builtin_fma(c[8], md, c[7]);
    j = c[9];
}


// The run tables of one plan x one group of <= 8 joints, built in LDS by the 256 lanes of the block together
// (32 lanes per joint: one per sampled switching time / candidate cut point / run). len = traj_len of the plan (> 0).
// Also applies the end-limit check of cc:59-61. Every thread of the block calls this.
// Orders the LDS traffic of the lanes of ONE wavefront (no s_barrier: the LDS serves a wave's requests in order).
LTP_DEV void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// What one lane contributes to the table build of a (plan, joint group) item, fetched ahead of time: lane k < 7 of a
// joint slot holds one switching time, lanes 7..12 the per-joint scalars; len / off are the plan's traj_len and packed
// offset (the same in every lane).
// With the table pass (k_build_tables) the lane instead holds up to kTableLoads 8-byte words of the finished tables:
// word (threadIdx.x >> 3) + 32 r of joint slot threadIdx.x & 7.
constexpr int kTableLoads = (kPackedWords + 31) / 32;
template <bool TABLES>
struct ItemRegs {
    int len;
    unsigned long long off;
    double pa, pb;
    unsigned long long w[kTableLoads];
};
template <>
struct ItemRegs<false> {
    int len;
    unsigned long long off;
    double pa, pb;
};

// global-memory form of the tables: lane index i = local plan * dof + joint; tiles of 64 lanes; inside a tile the 16-byte
// word PAIR (2k, 2k+1) of lane l sits at pair k, lane l — the table pass stores whole 1 KiB lines per wave, the joints of an
// item are neighbours, and a 16-byte-per-lane LDS-direct load (k_sample_tab's loader) fetches 64 consecutive pairs of one
// joint's table straight into the JointTable layout
LTP_DEV unsigned long long table_word_index(unsigned long long lane, int word)
{
    return (lane >> 6) * (unsigned long long)(kPackedWords * 64) + ((unsigned long long)(word >> 1) * 64ull + (lane & 63ull)) * 2ull + (unsigned long long)(word & 1);
}

// Issues the loads of an item (nothing here waits for them). p < 0: no item. tables != nullptr (TABLES): plan p is local
// plan p - tab_first of the table pass.
template <bool TABLES>
LTP_DEV ItemRegs<TABLES> fetch_item(long long p, int j0, int nj, int dof, const Limits& lim, const Queries& in, const Records& rec,
                                    const unsigned long long* __restrict__ offsets,
                                    const unsigned long long* __restrict__ tables = nullptr, long long tab_first = 0)
{
    ItemRegs<TABLES> r;
    r.len = 0; r.off = 0ull; r.pa = 0.0; r.pb = 0.0;
    if constexpr (TABLES) {
#pragma unroll
        for (int x = 0; x < kTableLoads; ++x) r.w[x] = 0ull;
    }
    if (p < 0) return r;
    r.len = rec.traj_len[p];
    if (offsets) r.off = offsets[p];
    if constexpr (TABLES) {
        const int jt = threadIdx.x & 7, wb = threadIdx.x >> 3;
        if (jt < nj) {
            const unsigned long long lane = (unsigned long long)(p - tab_first) * dof + j0 + jt;
#pragma unroll
            for (int x = 0; x < kTableLoads; ++x) {
                const int w = wb + 32 * x;
                if (w < kPackedWords) r.w[x] = tables[table_word_index(lane, w)];
            }
        }
        return r;
    }
    const int jl = threadIdx.x >> 5, k = threadIdx.x & 31;
    if (jl < nj) {
        const int j = j0 + jl;
        const long long rj = p * dof + j;
        const long long ix = p * in.sq + (long long)j * in.sj;
        if (k < 7) r.pa = rec.t_scaled[rj * 7 + k];
        else if (k == 7) { r.pa = rec.dir[rj]; r.pb = lim.j_max[j]; }
        else if (k == 8) { r.pa = rec.v_drive[rj]; r.pb = rec.dir[rj]; }
        else if (k == 9) r.pa = in.q_0[ix];
        else if (k == 10) r.pa = in.v_0[ix];
        else if (k == 11) r.pa = in.a_0[ix];
        else if (k == 12) r.pa = (double)rec.mod[rj];
    }
    return r;
}

// Leaves the run tables of the item in LDS. The caller must pass a block barrier before any wave reads them.
template <bool PROBE = false>
LTP_DEV void build_run_tables(SegTable& tab, long long p, int j0, int nj, int len, double Ts, const Limits& lim,
                              const Records& rec, double pa, double pb, unsigned long long* probe = nullptr)
{
    // ---- cooperative table build: thread t -> joint slot jl = t / 32, slot k = t % 32 ----
    // The 32 lanes of a joint slot sit in one wavefront and only ever exchange data with each other, so the steps
    // are separated by wave-level synchronisation (LDS operations of one wave complete in order); the whole block
    // meets once, at the end, before any wave reads another wave's tables.
    // The build is a short, latency-bound prologue that shares its SIMDs with other blocks' streaming waves:
    // give it issue priority, the bandwidth-bound streaming loop runs at the default priority.
    __builtin_amdgcn_s_setprio(3);
    const int jl = threadIdx.x >> 5, k = threadIdx.x & 31;
    const bool jact = jl < nj;
    const int j = j0 + (jact ? jl : 0);
    // (1) lane k < 7: one switching time each -> sampled index, lost fraction; lanes 7..12: per-joint scalars
    if (jact && k < 7) {
        const double tk = pa;
        const double fr = tk - Ts * dfloor(tk / Ts);                                   // cc:747
        tab.w.fr[jl][k] = fr;
        tab.w.frts[jl][k] = fr / Ts;
        tab.w.s[jl][k] = (k & 1) ? (int)dceil(tk / Ts) : (int)dfloor(tk / Ts);         // cc:751-757
    } else if (jact && k < 13) {
        // misc: dir*j_max, v_drive*dir (cc:823), q_0, v_0, a_0, mod
        tab.w.misc[jl][k - 7] = k < 9 ? pa * pb : pa;
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[3] = wall_clock64(); }
    // (2) lane 8: phase jerks and the nine correction terms of the joint; lanes < 20: one candidate cut point each
    int cval = -1;
    if (jact && k == 8) {
        const double dj = tab.w.misc[jl][0];
        const bool modp = tab.w.misc[jl][5] == 1.0;
        // cc:735-744: profile {1,0,-1,0,-1,0,1}, or {-1,0,1,0,-1,0,1} for the modified profile
        const double J0 = dj * (modp ? -1.0 : 1.0), J2 = dj * (modp ? 1.0 : -1.0), J4 = dj * -1.0, J6 = dj * 1.0;
        tab.w.Jp[jl][0] = J0; tab.w.Jp[jl][1] = dj * 0.0; tab.w.Jp[jl][2] = J2; tab.w.Jp[jl][3] = dj * 0.0;
        tab.w.Jp[jl][4] = J4; tab.w.Jp[jl][5] = dj * 0.0; tab.w.Jp[jl][6] = J6;
        const double* ft = tab.w.frts[jl];
        const double d20 = (tab.w.fr[jl][2] - tab.w.fr[jl][0]) / Ts;
        tab.w.corr[jl][0] = ft[0] * J0;                                   // j[s0+1]   cc:771
        tab.w.corr[jl][1] = (1 - ft[1]) * J2;                             // j[s1]     cc:773
        tab.w.corr[jl][2] = ft[2] * J2;                                   // j[s2+1]   cc:776
        tab.w.corr[jl][3] = d20 * J2;                                     // j[s1]     cc:781 (phase 2 absent): [0] then [3]
        tab.w.corr[jl][4] = (1 - ft[3]) * J4;                             // j[s3]     cc:787
        tab.w.corr[jl][5] = ft[4] * J4;                                   // j[s4+1]   cc:793
        tab.w.corr[jl][6] = 0.0;                                          // (cc:798, phases 2, 3 absent: [5], [0], [3] one by one)
        tab.w.corr[jl][7] = (1 - ft[5]) * J6;                             // j[s5]     cc:804
        tab.w.corr[jl][8] = ft[6] * J6;                                   // j[s6+1]   cc:807
    }
    if (jact && k < kCutSlots) {
        const int c = k == 0 ? 0 : tab.w.s[jl][kCutBase[k]] + kCutDelta[k];
        cval = (k == 0 || (c > 0 && c < len)) ? c : -1;
        tab.w.cand[jl][k] = cval;
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[4] = wall_clock64(); }
    // (3) sort + unique by counting: drop duplicates, then position = number of distinct valid values below
    if (jact && k < kCutSlots) {
        bool first = cval >= 0;
#pragma unroll
        for (int m = 0; m < kCutSlots; ++m) first = first && (m >= k || tab.w.cand[jl][m] != cval);   // fixed trip count: loads pipeline
        tab.w.runMode[jl][k] = first ? cval : -1;   // runMode doubles as scratch until step (4)
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[5] = wall_clock64(); }
    if (jact && k < kCutSlots) {
        const bool mine = tab.w.runMode[jl][k] >= 0;
        int pos = 0, distinct = 0;
#pragma unroll
        for (int m = 0; m < kCutSlots; ++m) {
            const int cm = tab.w.runMode[jl][m];
            if (cm >= 0) { ++distinct; if (cm < cval) ++pos; }
        }
        if (mine) tab.jt[jl].start[pos] = cval;
        if (k == 0) { tab.jt[jl].start[distinct] = len; tab.jt[jl].nseg = distinct; }
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[6] = wall_clock64(); }
    // (4) lane k < ns: mode, jerk and length of run k (the length parked in tab.jt[.].c[k][0] until step (6) overwrites it
    //     with the coefficients)
    const int ns = jact ? tab.jt[jl].nseg : 0;
    if (k < ns) {
        const int b = tab.jt[jl].start[k];
        const int* sj = tab.w.s[jl];
        const bool phase4 = sj[3] - sj[2] > 2;                                         // cc:813
        int mode = 0;
        if (b > sj[6]) mode |= kModeTail;
        if (phase4 && b >= sj[2] + 1 && b < sj[3] - 1) mode |= kModeVSnap;
        const double J = jerk_at(sj, tab.w.Jp[jl], tab.w.corr[jl], b);
        tab.w.runMode[jl][k] = mode;
        tab.w.runJ[jl][k] = J;
        // samples in the run (as an int in the low half of pre[0]: step (5) evaluates the run at its last sample)
        reinterpret_cast<int*>(tab.jt[jl].c[k])[0] = tab.jt[jl].start[k + 1] - b;
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[7] = wall_clock64(); }
    // (5) lane 0 of the joint: the state before each run — the only serial part. Each step is run_eval(run_coef(..))
    //     at the run's last sample, i.e. exactly what the streaming loop will store there (same functions, same bits).
    if (jact && k == 0) {
        const double vsnap = tab.w.misc[jl][1];
        double q = tab.w.misc[jl][2], v = tab.w.misc[jl][3], a = tab.w.misc[jl][4];   // state "before sample 0" (cc:810-812)
        // software-pipelined by hand: the next run's mode, jerk and length are fetched from LDS while the dependent chain of
        // the current run executes (the chain is a handful of binary64 operations, an LDS round trip is longer)
        int mode = tab.w.runMode[jl][0], cnt = reinterpret_cast<const int*>(tab.jt[jl].c[0])[0];
        double J = tab.w.runJ[jl][0];
        for (int m = 0; m < ns; ++m) {
            const int mn = m + 1 < ns ? m + 1 : m;
            const int mode_n = tab.w.runMode[jl][mn], cnt_n = reinterpret_cast<const int*>(tab.jt[jl].c[mn])[0];
            const double J_n = tab.w.runJ[jl][mn];
            tab.w.state[jl][m][0] = a; tab.w.state[jl][m][1] = v; tab.w.state[jl][m][2] = q;
            const RunCoef rc = run_coef(mode, J, a, v, q, vsnap, Ts);
            double jj;
            run_eval(rc.c, cnt, q, v, a, jj);
            mode = mode_n; cnt = cnt_n; J = J_n;
        }
        // cc:59-61: q now holds sample len-1
        if (q < lim.q_min[j] || q > lim.q_max[j]) atomicOr(&rec.status[p], kStatusEndLimit);
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[8] = wall_clock64(); }
    // (6) lane k < ns: the coefficients of run k
    if (k < ns) {
        const RunCoef rc = run_coef(tab.w.runMode[jl][k], tab.w.runJ[jl][k], tab.w.state[jl][k][0], tab.w.state[jl][k][1],
                                    tab.w.state[jl][k][2], tab.w.misc[jl][1], Ts);
#pragma unroll
        for (int x = 0; x < kRunCoefs; ++x) tab.jt[jl].c[k][x] = rc.c[x];
    }
    __builtin_amdgcn_s_setprio(0);
}

// Table-pass form of build_run_tables: the packed tables arrive in registers (fetch_item<true>), are placed at the end of their
// JointTable and expanded in place. Every thread of the block calls this; it ends with the block barrier after which any wave may
// read the tables.
LTP_DEV void install_run_tables(SegTable& tab, int nj, const unsigned long long (&w)[kTableLoads], double Ts)
{
    static_assert(kSampleJointGroup * kMaxSegments <= kSampleThreads && kSampleJointGroup * (kPackedHeaderWords - 2) <= kSampleThreads, "one task per thread");
    {
        const int jt = threadIdx.x & 7, wb = threadIdx.x >> 3;
        if (jt < nj) {
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(&tab.jt[jt]) + kPackedAt;
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
    if (hx < nj) hdr = reinterpret_cast<const unsigned long long*>(&tab.jt[hx])[kPackedAt + hw];
    bool live = false;
    RunCoef rc;
    if (jx < nj) {
        const unsigned long long* pk = reinterpret_cast<const unsigned long long*>(&tab.jt[jx]) + kPackedAt;
        const int nseg = (int)(unsigned)pk[0];
        if (r < nseg) {
            live = true;
            const double* st = reinterpret_cast<const double*>(pk + kPackedHeaderWords + r * kPackedRunWords);
            rc = run_coef<kSemMatlab>((int)(unsigned)pk[kPackedHeaderWords + r * kPackedRunWords + 4], st[3], st[0], st[1], st[2],
                                      reinterpret_cast<const double*>(pk)[12], Ts);   // (a superset of the C++ modes: same bits)
        }
    }
    __syncthreads();
    if (hx < nj) reinterpret_cast<unsigned long long*>(&tab.jt[hx])[hw] = hdr;
    if (live) {
#pragma unroll
        for (int x = 0; x < kRunCoefs; ++x) tab.jt[jx].c[r][x] = rc.c[x];
    }
    __syncthreads();
}

// Streams the rows of one item (plan x joint group) from the run tables in LDS. Every thread of the block calls this.
// (Pass B and pass A are written out in place: composed from two helper functions k_sample needed 96 instead of 89 VGPRs —
// one more than its budget of 5 blocks per CU allows — and the spill reload sat behind the look-ahead loads. tab_stream()
// below carries the same slot arithmetic for the table-pass sampler; tests/test_gpu_edge.py compares the rows of the two bit
// for bit.)
template <bool STREAMING, bool DRY, typename T>
LTP_DEV void stream_rows(SegTable& tab, int j0, int nj, int dof, int slen, unsigned long long stride, T* __restrict__ plan_base,
                         RowSpec rows)
{
    // Per joint, every lane produces q, v, a and j of N consecutive samples (a "slot": N = 2 doubles or 4 floats)
    // and issues four 16-B stores, i.e. four 1 KiB wave stores into the four rows of that joint. (Measured on
    // MI355X: for float64 rows this runs at the same rate as the identical store pattern without any arithmetic;
    // deeper unrolling, writing the rows one after the other, and walking the (joint, slot) space as one flat sequence
    // so that no step has idle lanes are all slower, the last one by 15 %.) float rows hold the binary64 results
    // rounded once.
    typedef typename OutVec<T>::type V;
    constexpr int N = OutVec<T>::N;
    const unsigned long long arr_stride = (unsigned long long)dof * stride;   // distance between q, v, a, j blocks
    const int nslots = (slen + N - 1) / N;
    const int sstride = rows.stride > 1 ? rows.stride : 1;

    // Pass B, once per item: the slots that contain a run boundary (and the row's last slot if it is partly padding).
    // There are at most 19 per row, but in the
    // row-by-row loop below most 64-slot wave steps contain one, and a wave that has one would execute the per-sample
    // path for all its lanes. So lane k of joint slot jl (the mapping of the table build) evaluates the slot of
    // boundary k, if that slot really straddles it and boundary k-1 has not claimed the same slot, and parks the four
    // 16-byte results in LDS (in the space of the build scratch); the main loop picks them up, so that it still
    // writes every row as full contiguous wave stores (leaving holes for scattered 16-byte stores costs 13 % of the
    // float64 bandwidth).
    if constexpr (!DRY) {
        const int jl = threadIdx.x >> 5, k = threadIdx.x & 31;
        const int nruns = jl < nj ? tab.jt[jl].nseg : 0;
        if (k < nruns) {
            const int* st = tab.jt[jl].start;
            // lane k >= 1: the slot of boundary k; lane 0: the last slot of the row if the row ends inside it (its
            // tail is padding), so that the main loop never has to mask anything
            const int u = k >= 1 ? (st[k] + sstride - 1) / sstride      // first stored sample at or after boundary k
                                 : slen;
            bool mine = (u % N) != 0 && u < N * nslots;
            if (mine && k > 1) {
                const int up = (st[k - 1] + sstride - 1) / sstride;
                if ((up % N) != 0 && up / N == u / N) mine = false;     // boundary k-1 owns this slot
            }
            if (mine) {
                const int i0 = u / N * N, t0 = i0 * sstride;
                int kh = k >= 1 ? k - 1 : nruns - 1;
                while (st[kh] > t0) --kh;                               // run of the slot's first sample (st[0] = 0)
                int ch = st[kh], nh = kh + 1 < nruns ? st[kh + 1] : 0x7fffffff;
                V o[4];
#pragma unroll
                for (int h = 0; h < N; ++h) {
                    const int i = t0 + h * sstride;
                    while (nh <= i) {
                        ++kh;
                        ch = nh;
                        nh = kh + 1 < nruns ? st[kh + 1] : 0x7fffffff;
                    }
                    const bool pad = i0 + h >= slen;                    // the tail of the last slot is row padding
                    double x4[4];
                    run_eval(tab.jt[jl].c[kh], i - ch + 1, x4[0], x4[1], x4[2], x4[3]);
#pragma unroll
                    for (int x = 0; x < 4; ++x) o[x][h] = pad ? (T)0 : (T)x4[x];
                }
#pragma unroll
                for (int x = 0; x < 4; ++x) *reinterpret_cast<V*>(&tab.bnd[jl][k][x]) = o[x];
            }
        }
        __syncthreads();
    }

    // Pass A: row by row; the N samples of any other slot lie in one run, whose coefficients are read once
    // Rows shorter than the block (first-N-samples rows) are shared out so that no wave idles: wpr waves per row,
    // 4 / wpr rows at a time. Long rows: wpr = 4, i.e. all 256 lanes on one row after the other.
    const int lw = nslots <= 64 ? 0 : (nslots <= 128 ? 1 : 2);                      // wpr = 1 << lw
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    // Streaming float64 rows: buffer stores through descriptors of the four rows being written (base and size in SGPRs,
    // one 32-bit lane offset for all four stores, anything beyond the row dropped by the hardware's range check),
    // non-temporal at agent scope ("sc1 nt"; only the buffer builtins take the cache-policy bits). Measured on MI355X
    // against the compiler's non-temporal global store, same box: +0.5-0.9 % for float64 rows (7.06 -> 7.09, 6.97 -> 7.02
    // TB/s), but -3 % for float32 rows, which therefore keep the global store. A descriptor spans at most 1 GiB, so a
    // longer row — 64 M float64 samples — is written window by window; any realistic row is one window.
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    constexpr int kWindowSlots = 1 << 26;
    constexpr bool kBufferStores = STREAMING && sizeof(T) == 8;
    for (int jl2 = wave >> lw; jl2 < nj; jl2 += 4 >> lw) {
        T* const row = plan_base + (unsigned long long)(j0 + jl2) * stride;
        const int* st = tab.jt[jl2].start;
        const int nruns = tab.jt[jl2].nseg;
        // run cursor of this lane: samples [cur, nxt) belong to run kr (nxt = INT_MAX for the last run)
        int kr = 0, cur = 0, nxt = nruns > 1 ? st[1] : 0x7fffffff;
        for (int wbase = 0; wbase < nslots; wbase += kWindowSlots) {
            const int wend = nslots - wbase < kWindowSlots ? nslots : wbase + kWindowSlots;
            __amdgpu_buffer_rsrc_t rsrc[4];
            if constexpr (kBufferStores) {
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    // (plan_base and everything else in this address is wave-uniform: scalar arithmetic)
                    rsrc[x] = __builtin_amdgcn_make_buffer_rsrc(row + x * arr_stride + (unsigned long long)wbase * N, 0,
                                                                (wend - wbase) * (int)sizeof(V), 0x00020000);
                }
            }
            for (int slot = wbase + ((wave & ((1 << lw) - 1)) << 6) + lane; slot < wend; slot += 64 << lw) {
                const int i0 = N * slot;              // first stored sample of this slot; it is sample i0*sstride of the trajectory
                V o[4];
                if constexpr (DRY) {
#pragma unroll
                    for (int x = 0; x < 4; ++x)
#pragma unroll
                        for (int h = 0; h < N; ++h) o[x][h] = (T)(i0 + h);
                } else {
                    const int t0 = i0 * sstride;
                    while (nxt <= t0) {
                        ++kr;
                        cur = nxt;
                        nxt = kr + 1 < nruns ? st[kr + 1] : 0x7fffffff;
                    }
                    const bool straddles = t0 + (N - 1) * sstride >= nxt;
                    if (straddles || i0 + N > slen) {
                        // run boundary kr+1 lies inside the slot, or the row ends inside it: pass B has left the
                        // finished values in LDS (entry 0 is the row's last slot)
                        const int e = straddles ? kr + 1 : 0;
#pragma unroll
                        for (int x = 0; x < 4; ++x) o[x] = *reinterpret_cast<const V*>(&tab.bnd[jl2][e][x]);
                    } else {
                        double c[kRunCoefs];
#pragma unroll
                        for (int x = 0; x < kRunCoefs; ++x) c[x] = tab.jt[jl2].c[kr][x];
#pragma unroll
                        for (int h = 0; h < N; ++h) {
                            double x4[4];
                            run_eval(c, t0 + h * sstride - cur + 1, x4[0], x4[1], x4[2], x4[3]);
#pragma unroll
                            for (int x = 0; x < 4; ++x) o[x][h] = (T)x4[x];
                        }
                    }
                }
                if constexpr (kBufferStores) {
                    const unsigned voff = (unsigned)(slot - wbase) * (unsigned)sizeof(V);
#pragma unroll
                    for (int x = 0; x < 4; ++x)
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[x]), rsrc[x], voff, 0, /*nt | sc1*/ 2 | 16);
                } else if constexpr (STREAMING) {
#pragma unroll
                    for (int x = 0; x < 4; ++x) __builtin_nontemporal_store(o[x], reinterpret_cast<V*>(row + x * arr_stride + i0));
                } else {
#pragma unroll
                    for (int x = 0; x < 4; ++x) *reinterpret_cast<V*>(row + x * arr_stride + i0) = o[x];
                }
            }
        }
    }
}

// Persistent work-queue form: as many blocks as the chip holds, each pulling (plan, joint group) items from one
// counter until it runs dry. With static round-robin dispatch the eight XCDs finish their equal shares up to 15 %
// apart (they do not write to all HBM channels at the same speed), which left a 2-4 ms tail of a 23-29 ms launch at
// reduced bandwidth; pulling keeps every XCD busy to the end. The counter sees ~20 pulls/us, far below the
// ~90/us a single word sustains. Exit: every block leaves as soon as it draws an item >= total.
// Item order: item i -> plan (i % spread) * ceil(count/spread) + i / spread (spread = 64 by default), so blocks that
// are resident together write all over the output tile: on MI355X a narrow moving write front only reaches
// ~5.2 TB/s while writes spread over a large tile reach the fill-kernel ceiling ("What bounds the sampler").
// float32 rows need 4 samples per lane in flight: they get the register budget of 4 blocks per CU (with 5 the
// compiler spills, and a spill reload waits for every outstanding load of the wave, i.e. for the look-ahead).
// The loop runs one item ahead: the next item is drawn while the tables of the current one are being built, and its
// records are requested just before the current item's rows are streamed, so that the three dependent round trips
// an item needs (queue counter, traj_len / offset, records) travel under ~10^2 row stores instead of in front of them.
// (On gfx950 a wave's loads and stores share one in-order counter: a load result consumed behind a run of stores
// waits for all of them, so an item pays one drain of its own stores either way — but only one.)
template <bool STREAMING, bool DRY, typename T>
__global__ void __launch_bounds__(kSampleThreads, (sizeof(T) == 4 ? kSampleBlocksPerCU - 1 : kSampleBlocksPerCU))
k_sample(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
         const unsigned long long* __restrict__ offsets, T* __restrict__ out, unsigned long long capacity,
         unsigned long long* __restrict__ stamps, int spread, RowSpec rows, unsigned long long* __restrict__ next_item,
         int draw_chunk /* items per queue draw, a power of two */)
{
    __shared__ SegTable tab;
    __shared__ unsigned long long s_item;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    const long long per = (count + spread - 1) / spread;
    const unsigned long long total = (unsigned long long)per * spread * ngroups;
    const unsigned long long off0 = offsets[first];
    // Queue draws. One device-scope counter sustains ~90 atomics/us; items of short (capped) rows are drawn faster than
    // that, so a draw takes draw_chunk consecutive items (1 for whole trajectories: a long item at the end of a launch is
    // a long tail). Thread 0 keeps the chunk; chunk_i is the same in every thread.
    unsigned long long chunk_base = 0ull;
    int chunk_i = 0;
    auto draw = [&]() -> unsigned long long {      // thread 0 only
        if (chunk_i == 0) chunk_base = atomicAdd(next_item, (unsigned long long)draw_chunk);
        return chunk_base + (unsigned long long)chunk_i;
    };

    // item -> (local plan, joint group); local >= count are the holes of the interleave
    auto decode = [&](unsigned long long item, long long& local, int& j0, int& nj) {
        const int group = (int)(item % ngroups);
        const long long slot = (long long)(item / ngroups);
        local = (slot % spread) * per + slot / spread;
        j0 = group * kSampleJointGroup;
        nj = (dof - j0) < kSampleJointGroup ? (dof - j0) : kSampleJointGroup;
    };
    auto fetch = [&](unsigned long long item) {
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        const bool some = item < total && local < count;
        return fetch_item<false>(some ? first + local : -1, j0, nj, dof, lim, in, rec, offsets);
    };

    if (threadIdx.x == 0) s_item = draw();
    chunk_i = (chunk_i + 1) & (draw_chunk - 1);
    __syncthreads();
    unsigned long long item = s_item;
    ItemRegs<false> cur = fetch(item);
    __syncthreads();   // s_item may be rewritten
    while (item < total) {
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        const long long p = first + local;
        const bool lead = threadIdx.x == 0 && j0 == 0;
        // diagnostic only (stamps == nullptr in every product call): start / tables ready / end on the 100 MHz wall clock
        if (stamps && lead && local < count) stamps[3 * local] = wall_clock64();
        // the same in every lane, but loaded per lane: readfirstlane moves them (and all the row addressing derived from
        // them) into scalar registers
        const int len = __builtin_amdgcn_readfirstlane(cur.len);   // 0: hole, failed or non-finite query -> nothing to sample
        const unsigned long long off = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur.off >> 32)) << 32) |
                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur.off);
        const unsigned long long rel = off - off0;
        const int slen = stored_len(len, rows);           // samples actually stored per row
        const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
        bool ok = len > 0;
        if (ok && rel + 4ull * dof * stride > capacity) {
            if (lead) atomicOr(&rec.status[p], kStatusOverflow);
            ok = false;
        }
        unsigned long long drawn = 0ull;
        if (threadIdx.x == 0) drawn = draw();                          // an atomic returns while the tables are built
        chunk_i = (chunk_i + 1) & (draw_chunk - 1);
        if (ok) build_run_tables(tab, p, j0, nj, len, t_sample, lim, rec, cur.pa, cur.pb);
        if (threadIdx.x == 0) s_item = drawn;
        __syncthreads();                                               // tables complete, next item known
        const unsigned long long nitem = s_item;
        const ItemRegs<false> nxt = fetch(nitem);                      // in flight while this item streams
        if (ok) {
            if (stamps && lead) stamps[3 * local + 1] = wall_clock64();   // run tables ready
            stream_rows<STREAMING, DRY, T>(tab, j0, nj, dof, slen, stride, out + rel, rows);
        }
        __syncthreads();                                               // tables and s_item are free again
        if (stamps && ok && lead) stamps[3 * local + 2] = wall_clock64();
        item = nitem;
        cur = nxt;
    }
}

// ---------------------------------------------------------------------------------------
// The sampler for rows that are short compared with an item's fixed costs (first-N-samples rows, receding-horizon rows).
// k_sample pays per item: a table build of ~8 us of latency, three block barriers, and — because a wave's loads and stores
// share one in-order counter on gfx950 — one complete drain of the wave's own row stores before it can consume the next
// item's prefetched records. For 1.7 k-sample rows other blocks of the CU cover that; for a few hundred samples per row it
// is most of the item (measured: 3.6-3.8 TB/s for 256-sample rows, the same with and without the table build).
// Here the roles are split between the waves of a block:
//   * the run tables come from the table pass (k_build_tables), compact: only the runs the stored samples touch;
//   * the last wave, the loader, draws the coming items and brings their tables into one of kTabBuffers LDS buffers with
//     LDS-direct loads, two items ahead; it issues loads but never row stores;
//   * the other waves, the streaming waves, each own one joint of the current item and write its rows from the item's
//     buffer, slots with a run boundary evaluated in place; they issue stores but never loads, so nothing they execute ever
//     waits for a store to complete;
//   * buffers change hands through LDS flags (s_ready / s_consumed in sample_tab_body), not block barriers: a fast wave
//     runs up to kTabBuffers - 1 items ahead of a slow one.
// Rows are bit-identical to k_sample's: same tables (for_each_run == the cooperative build), same per-sample arithmetic.
// ---------------------------------------------------------------------------------------
constexpr int kTabStreamWaves = 7;                              // streaming waves per block; wave kTabStreamWaves is the loader
constexpr int kTabThreads = (kTabStreamWaves + 1) * 64;          // 8 waves: two per SIMD
constexpr int kTabJointGroup = 7;                                // joints per item: one row set per streaming wave
constexpr int kTabBuffers = 4;                                   // LDS table buffers per block (11.9 KB each)
struct TabItem {
    unsigned long long rel;               // element offset of the plan inside `out`
    int slen;                             // stored samples per row; 0 = nothing to stream (hole, failed plan, does not fit)
    int j0, nj;
    int done;                             // 1 = the queue is exhausted
    unsigned long long item;              // queue position (diagnostic stamps only)
};
struct alignas(16) TabBuffer {
    JointTable jt[kTabJointGroup];        // filled by LDS-direct loads (16 bytes per lane) from the table pass's output
    TabItem hdr;                          // what the streaming waves read, written by the loader when the loads are in
};
static_assert(sizeof(JointTable) % 16 == 0, "LDS-direct loads land 16 bytes per lane");

// What a streaming wave does with an item: wave w owns joint w (nj <= 3: several waves share a joint) and writes the
// joint's four rows, 64 slots = 1 KiB per row and step. Same slot arithmetic as stream_rows' pass A, except that a slot which
// contains a run boundary (or the end of the row) is evaluated sample by sample in place, by exactly the steps of
// stream_rows' pass B, instead of being picked up from LDS: rows this kernel is meant for are one or two wave steps long, and a
// pass B costs the wave ~2 us per item in which it issues no store.
template <bool STREAMING, typename T>
LTP_DEV void tab_stream(const TabBuffer& B, const TabItem& hdr /* B.hdr, already in registers */, int dof, T* __restrict__ out, int sstride, int wave)
{
    typedef typename OutVec<T>::type V;
    constexpr int N = OutVec<T>::N;
    // the lane id is recomputed per item: kept in a register across the kernel it ends up spilled (the loader branch needs
    // the registers), and a scratch reload here would wait for every row store the wave has in flight
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int slen = hdr.slen, j0 = hdr.j0, nj = hdr.nj;
    if (slen <= 0) return;
    const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    const unsigned long long arr_stride = (unsigned long long)dof * stride;
    const int nslots = (slen + N - 1) / N;
    // Lanes per joint: a row of at most 32 (16) slots leaves half (three quarters) of a wave without a slot, and what a
    // streaming wave costs is the instructions it issues, not the lanes that execute them: such rows share a wave between
    // two (four) joints — 64-sample rows of a 7-joint item take four waves' worth of instructions instead of seven.
    const int lg = nslots > 32 ? 6 : (nslots > 16 ? 5 : 4);
    int jl2, first_slot, step;
    if (lg == 6) {
        const int wpr = nj >= 4 ? 1 : (nj == 3 ? 2 : (nj == 2 ? 3 : 7));     // waves per joint
        jl2 = wpr == 1 ? wave : (wpr == 2 ? wave >> 1 : (wpr == 3 ? (wave >= 3) + (wave >= 6) : 0));   // wave / wpr without a division
        first_slot = (wave - jl2 * wpr) * 64 + lane;
        step = 64 * wpr;
    } else {
        jl2 = (wave << (6 - lg)) + (lane >> lg);
        first_slot = lane & ((1 << lg) - 1);
        step = 64;                                                         // (one slot per lane)
    }
    const bool mine = jl2 < nj;
    if (__builtin_amdgcn_ballot_w64(mine) == 0ull) return;
    const JointTable& jt = B.jt[mine ? jl2 : 0];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // one buffer descriptor over the item's rows (all four arrays of all its joints), a 32-bit offset per lane; items beyond
    // 4 GB of rows (trajectories of millions of samples) take ordinary stores
    const unsigned long long item_bytes = 4ull * arr_stride * sizeof(T);
    const bool buffer_stores = STREAMING && sizeof(T) == 8 && item_bytes <= 0xffffff00ull;
    T* const item = out + hdr.rel;
    const unsigned long long row_at = (unsigned long long)(j0 + (mine ? jl2 : 0)) * stride;     // element offset of the joint's q row
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(item, 0, buffer_stores ? (int)(unsigned)item_bytes : 0, 0x00020000);
    const int* st = jt.start;
    const int nruns = jt.nseg;
    int kr = 0, cur = 0, nxt = nruns > 1 ? st[1] : 0x7fffffff;
    for (int slot = mine ? first_slot : nslots; slot < nslots; slot += step) {
        const int i0 = N * slot;
        V o[4];
        const int t0 = i0 * sstride;
        while (nxt <= t0) {
            ++kr;
            cur = nxt;
            nxt = kr + 1 < nruns ? st[kr + 1] : 0x7fffffff;
        }
        const bool straddles = t0 + (N - 1) * sstride >= nxt;
        if (straddles || i0 + N > slen) {
            // a run boundary or the end of the row inside the slot: sample by sample (the tail of the last slot is
            // row padding and stays zero)
            int kh = kr, ch = cur, nh = nxt;
#pragma unroll
            for (int h = 0; h < N; ++h) {
                const int i = t0 + h * sstride;
                while (nh <= i) {
                    ++kh;
                    ch = nh;
                    nh = kh + 1 < nruns ? st[kh + 1] : 0x7fffffff;
                }
                const bool pad = i0 + h >= slen;
                double x4[4];
                run_eval(jt.c[kh], i - ch + 1, x4[0], x4[1], x4[2], x4[3]);
#pragma unroll
                for (int x = 0; x < 4; ++x) o[x][h] = pad ? (T)0 : (T)x4[x];
            }
        } else {
            double c[kRunCoefs];
#pragma unroll
            for (int x = 0; x < kRunCoefs; ++x) c[x] = jt.c[kr][x];
#pragma unroll
            for (int h = 0; h < N; ++h) {
                double x4[4];
                run_eval(c, t0 + h * sstride - cur + 1, x4[0], x4[1], x4[2], x4[3]);
#pragma unroll
                for (int x = 0; x < 4; ++x) o[x][h] = (T)x4[x];
            }
        }
        if (buffer_stores) {
            if constexpr (STREAMING && sizeof(T) == 8) {
                const unsigned voff = (unsigned)((row_at + (unsigned long long)i0) * sizeof(T));
                const unsigned arr_bytes = (unsigned)(arr_stride * sizeof(T));
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[x]), rsrc, voff + (unsigned)x * arr_bytes, 0, /*nt | sc1*/ 2 | 16);
            }
        } else if constexpr (STREAMING) {
#pragma unroll
            for (int x = 0; x < 4; ++x) __builtin_nontemporal_store(o[x], reinterpret_cast<V*>(item + row_at + x * arr_stride + i0));
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) *reinterpret_cast<V*>(item + row_at + x * arr_stride + i0) = o[x];
        }
    }
}

// The lane id, recomputed (two instructions) and opaque to common-subexpression elimination: kept live across the loader's
// loop it gets spilled, and a scratch reload in that loop waits for the table loads in flight (~8 us each time).
LTP_DEV int fresh_lane()
{
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// s_waitcnt with only the vector-memory counter set (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
#define LTP_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt((((N) & 15) | (((N) >> 4) << 14)) | (7 << 4) | (15 << 8))

// LDS accesses of the loader wave, as instructions the compiler does not model. Once a wave has LDS-direct loads in flight
// the compiler puts "wait for ALL vector-memory operations" in front of every LDS access it knows of (it cannot tell that the
// loads land elsewhere), which would drain the prefetches at every flag poll and header write. These are plain ds_read /
// ds_write: LDS serves a wave's requests in order, reads wait for their own data, and the loader orders them against the
// LDS-direct loads itself (LTP_WAIT_VMCNT). Untracked LDS operations can only make the compiler's own lgkmcnt waits stricter.
LTP_DEV unsigned lds_offset(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; }
LTP_DEV int lds_peek32(unsigned a)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// several reads, one wait: an LDS round trip is ~150 cycles in a CU full of streaming waves
LTP_DEV void lds_peek64x4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned long long (&v)[4])
{
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
LTP_DEV void lds_peek64x5(unsigned a, unsigned long long (&v)[5])      // five consecutive 8-byte words
{
    asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:8\n\tds_read_b64 %2, %5 offset:16\n\tds_read_b64 %3, %5 offset:24\n\t"
                 "ds_read_b64 %4, %5 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]) : "v"(a) : "memory");
}
LTP_DEV void lds_poke32(unsigned a, int v) { asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory"); }
LTP_DEV void lds_poke64(unsigned a, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(v) : "memory"); }

// Loader wave: turns the packed tables that LDS-direct loads left at the end of the item's JointTables into the tables the
// streaming waves read (header words to the front, ten coefficients per run from its five packed words, by run_coef()).
// Lane -> (joint lane / 9, run 9 * pass + lane % 9): 63 lanes per pass, one pass for capped rows, three for whole tables.
// Order matters, because the expanded words overwrite the packed ones: a pass writes JointTable words 12 + 90 k .. 101 + 90 k,
// the packed header sits in words 98 .. 111 and the packed run r in words 112 + 5 r .. 116 + 5 r — so the header (and vsnap) is
// read before pass 0 writes, every lane of a pass reads its run before any lane writes (the LDS serves a wave in order and
// lds_peek waits for its data), and what pass k overwrites is below what the later passes still have to read.
// The same expansion by the streaming wave that owns the joint (ordinary LDS accesses: a streaming wave has no LDS-direct loads
// in flight). Lane r < nseg expands run r, lanes 32..43 move the header; every lane has read before any lane writes (one wave,
// one instruction stream, and the LDS serves it in order).
template <class Buffer>
LTP_DEV void expand_packed_tables(Buffer& B, int nj, int max_runs, int lane, double Ts)
{
    constexpr int kRunsPerPass = 9;
    static_assert(kTabJointGroup * kRunsPerPass <= 64, "a pass is one wave");
    static_assert(12 + kRunCoefs * kRunsPerPass <= kPackedAt + kPackedHeaderWords - 2, "pass 0 stays below vsnap and the packed runs");
    static_assert(12 + kRunCoefs * 2 * kRunsPerPass <= kPackedAt + kPackedHeaderWords + kPackedRunWords * 2 * kRunsPerPass && kMaxSegments <= 3 * kRunsPerPass,
                  "pass 1 stays below the packed runs of pass 2, and there is no pass 3");
    const int x = lane / kRunsPerPass, i = lane - x * kRunsPerPass;
    const bool joint = x < nj;
    const unsigned jt = lds_offset(&B.jt[joint ? x : 0]);
    const unsigned pk = jt + (unsigned)kPackedAt * 8u;
    unsigned long long hd[4];
    lds_peek64x4(pk + 8u * (unsigned)i, pk + 8u * (unsigned)(9 + (i < 3 ? i : 0)), pk, pk + 12u * 8u, hd);
    const int nseg = (int)(unsigned)hd[2];
    const double vsnap = __builtin_bit_cast(double, hd[3]);
    if (joint) {
        lds_poke64(jt + 8u * (unsigned)i, hd[0]);
        if (i < 3) lds_poke64(jt + 8u * (unsigned)(9 + i), hd[1]);
    }
    for (int r0 = 0; r0 < max_runs; r0 += kRunsPerPass) {
        const int r = r0 + i;
        const bool live = joint && r < nseg;
        const unsigned src = pk + (unsigned)(kPackedHeaderWords + kPackedRunWords * (live ? r : 0)) * 8u;
        unsigned long long st[5];
        lds_peek64x5(src, st);
        const RunCoef rc = run_coef<kSemMatlab>((int)(unsigned)st[4], __builtin_bit_cast(double, st[3]), __builtin_bit_cast(double, st[0]),
                                                __builtin_bit_cast(double, st[1]), __builtin_bit_cast(double, st[2]), vsnap, Ts);   // (a superset of the C++ modes: same bits)
        if (live) {
            const unsigned dst = jt + (unsigned)(12 + kRunCoefs * r) * 8u;
#pragma unroll
            for (int c = 0; c < kRunCoefs; ++c) lds_poke64(dst + 8u * (unsigned)c, __builtin_bit_cast(unsigned long long, rc.c[c]));
        }
    }
}

template <bool STREAMING, typename T>
LTP_DEV void sample_tab_body(long long first, long long count, long long base_first, int dof, Records rec,
                             const unsigned long long* __restrict__ offsets, T* __restrict__ out, unsigned long long capacity, int spread,
                             RowSpec rows, unsigned long long* __restrict__ next_item, const unsigned long long* __restrict__ tables,
                             int draw_chunk, unsigned long long* __restrict__ stamps /* diagnostic: 8 per item, nullptr in product calls */,
                             double t_sample)
{
    // stamps[8 * item + k] (wall clock, tools/tab_probe.py): loader — 0 its iteration starts (a buffer is free), 6 the next
    // item's loads are issued, 1 this item's loads are in, 3 it is published, 7 the previous publication, 2 = 1 if the tables
    // needed the second fetch; streaming wave 0 — 4 it starts the item, 5 its rows are issued.
    // Hand-over of the table buffers without block barriers: buffer s % kTabBuffers holds the block's s-th item once the
    // loader has set s_ready[s % kTabBuffers] = s + 1; streaming wave w has finished s_consumed[w] items. The loader reuses
    // a buffer when every wave is past the item that was in it; a fast wave may thus run kTabBuffers - 1 items ahead of a
    // slow one (with one barrier per item every wave waited for the slowest: 3.1 of 8.8 us at 256-sample rows). All eight
    // waves of a block are resident together, the loader waits only for the streaming waves and they only for the loader, the
    // loader publishes a final "done" item and every wave leaves on reading it: no wait can last forever.
    __shared__ TabBuffer buf[kTabBuffers];
    __shared__ int s_ready[kTabBuffers];
    __shared__ int s_consumed[kTabStreamWaves];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sstride = rows.stride > 1 ? rows.stride : 1;
    if (threadIdx.x < kTabBuffers) s_ready[threadIdx.x] = 0;
    if (threadIdx.x < kTabStreamWaves) s_consumed[threadIdx.x] = 0;
    __syncthreads();
    if (wave < kTabStreamWaves) {
        // ---- streaming waves: LDS reads and row stores only ----
        for (int seq = 0;; ++seq) {
            const int b = seq % kTabBuffers;
            while (__hip_atomic_load(&s_ready[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq + 1) __builtin_amdgcn_s_sleep(1);   // (longer sleeps: no gain)
            // the header in one LDS round trip (two 16-byte reads), wave-uniform
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            static_assert(sizeof(TabItem) == 32 && offsetof(TabItem, slen) == 8 && offsetof(TabItem, nj) == 16 && offsetof(TabBuffer, hdr) % 16 == 0, "read as two u32x4");
            const u32x4 h0 = reinterpret_cast<const u32x4*>(&buf[b].hdr)[0], h1 = reinterpret_cast<const u32x4*>(&buf[b].hdr)[1];
            if (__builtin_amdgcn_readfirstlane((int)h1[1])) break;                       // done
            TabItem hdr;
            hdr.rel = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)h0[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)h0[0]);
            hdr.slen = __builtin_amdgcn_readfirstlane((int)h0[2]);
            hdr.j0 = __builtin_amdgcn_readfirstlane((int)h0[3]);
            hdr.nj = __builtin_amdgcn_readfirstlane((int)h1[0]);
            const bool stamp = stamps && wave == 0 && (threadIdx.x & 63) == 0;
            const unsigned long long it = stamp ? ((unsigned long long)h1[3] << 32) | h1[2] : 0ull;
            if (stamp) stamps[8 * it + 4] = wall_clock64();
            tab_stream<STREAMING, T>(buf[b], hdr, dof, out, sstride, wave);
            if (stamp) stamps[8 * it + 5] = wall_clock64();
            // the wave's reads of buf[b] are complete (release orders its LDS traffic; row stores need not be: they carry
            // their data in registers)
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&s_consumed[wave], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // ---- loader wave: loads only. Under the sampler's own write traffic a global read takes ~8 us, longer than an item
    // streams, so an item's data is requested kTabAhead items before it is published — and it is requested with LDS-DIRECT
    // loads (buffer_load ... lds, 16 bytes per lane straight into the JointTable of the item's buffer): no registers are held,
    // and because such a load has no register result the compiler inserts no wait of its own for it; the one wait is written
    // here. Loads complete in order, and every item issues exactly kPerItem of them (holes and the items after the end of the
    // queue re-read item 0's tables into a buffer nobody streams), so "item seq is in" is "at most kTabAhead * kPerItem
    // vector-memory operations outstanding" — a constant. Anything else the loader issues in between (queue draws, status
    // bits, stamps) only makes that wait stricter. The loader shares its SIMD with five streaming waves that keep the vector
    // ALU busy: it runs at raised issue priority. ----
    __builtin_amdgcn_s_setprio(3);
    constexpr int kTabAhead = 2;
    constexpr int kCappedRuns = 8;
    constexpr int kCappedPairs = (kPackedHeaderWords + kCappedRuns * kPackedRunWords) / 2;   // 27 word pairs: header + 8 packed runs
    constexpr unsigned kPackedByte = (unsigned)kPackedAt * 8u;                               // where the packed form lands in a JointTable
    static_assert(kTabAhead + 2 <= kTabBuffers, "buffers: one being streamed, one being published, kTabAhead in flight");
    const int ngroups = (dof + kTabJointGroup - 1) / kTabJointGroup;
    const long long per = (count + spread - 1) / spread;
    const unsigned long long total = (unsigned long long)per * spread * ngroups;
    constexpr unsigned kTileBytes = (unsigned)kPackedWords * 64u * 8u;
    // capped rows mostly touch the first runs only (a switch of the jerk profile cuts up to three runs: 8 runs is what
    // 256 samples of a 7-DoF plan need in 99.95 % of the items): kCappedPairs word pairs (54 words, 8 packed runs) per joint — an
    // item's joints are neighbours in the table tile, so its seven loads fetch 27 lines of 128 bytes; whole tables are 57 pairs
    const bool whole_tables = rows.max_samples <= 0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto uniform64 = [](unsigned long long x) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
    };
    // Queue positions: a draw takes draw_chunk consecutive items (one device-scope counter sustains ~90 atomics/us; short
    // items are drawn faster than that). The atomic of the next chunk is issued at the top of the iteration that hands out
    // the current chunk's last position and consumed at the bottom of the SAME iteration: in between lies straight-line code,
    // so the compiler can count what was issued after it and waits for the atomic alone — carried across the loop it would
    // wait for everything outstanding, i.e. drain the prefetches.
    unsigned long long chunk_cur = 0ull, chunk_pending = 0ull;
    int chunk_i = 0;
    bool chunk_wanted = false;
    if (fresh_lane() == 0) chunk_pending = atomicAdd(next_item, (unsigned long long)draw_chunk);
    chunk_cur = uniform64(chunk_pending);
    auto next_item_id = [&]() __attribute__((always_inline)) -> unsigned long long {       // top of an iteration
        const unsigned long long id = chunk_cur + (unsigned long long)chunk_i;
        if (++chunk_i == draw_chunk) {
            if (fresh_lane() == 0) chunk_pending = atomicAdd(next_item, (unsigned long long)draw_chunk);
            chunk_wanted = true;
        }
        return id;
    };
    auto finish_draw = [&]() __attribute__((always_inline)) {                               // bottom of the same iteration
        __builtin_amdgcn_sched_barrier(0);              // not to be hoisted in front of the iteration's loads (it would wait there)
        if (chunk_wanted) {
            chunk_cur = uniform64(chunk_pending);
            chunk_i = 0;
            chunk_wanted = false;
        }
    };
    // item -> (plan inside [first, first + count) or -1, first joint, joints). 64-bit divisions are ~10^2 instructions each on
    // this machine and the loader pays them per item: one joint group (dof <= 7) and a power-of-two interleave (the default,
    // 64) need none, anything else that fits 32 bits uses 32-bit division.
    const int spread_log2 = (spread & (spread - 1)) == 0 ? 31 - __builtin_clz((unsigned)spread) : -1;
    auto decode = [&](unsigned long long item, long long& local, int& j0, int& nj) __attribute__((always_inline)) {
        local = -1; j0 = 0; nj = 0;
        if (item >= total) return;
        unsigned long long slot = item;
        int group = 0;
        if (ngroups > 1) {
            if (total <= 0xffffffffull) { group = (int)((unsigned)item % (unsigned)ngroups); slot = (unsigned)item / (unsigned)ngroups; }
            else { group = (int)(item % (unsigned long long)ngroups); slot = item / (unsigned long long)ngroups; }
        }
        long long l;
        if (spread_log2 >= 0) l = (long long)(slot & (unsigned long long)(spread - 1)) * per + (long long)(slot >> spread_log2);
        else if (total <= 0xffffffffull) l = (long long)((unsigned)slot % (unsigned)spread) * per + (long long)((unsigned)slot / (unsigned)spread);
        else l = (long long)(slot % (unsigned long long)spread) * per + (long long)(slot / (unsigned long long)spread);
        j0 = group * kTabJointGroup;
        nj = (dof - j0) < kTabJointGroup ? (dof - j0) : kTabJointGroup;
        if (l < count) local = l;
    };
    // issues the kPerItem loads of an item into buffer B (nothing here waits)
    auto request = [&](TabBuffer& B, unsigned long long item) __attribute__((always_inline)) {
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        const bool real = local >= 0;
        const int lane = fresh_lane();
        // (the plan's length and row offset come with its tables: JointTable::len, start[kMaxSegments + 1]; holes re-read plan `first`)
        // tables: per joint slot one load of up to 57 word pairs (the packed form, landing at the end of the JointTable it is
        // expanded into); a descriptor over the two tiles the item's joints can lie in, one 32-bit offset per lane
        const unsigned long long l0 = (unsigned long long)(real ? local : 0) * dof + (real ? j0 : 0);       // wave-uniform
        const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned long long*>(tables) + (l0 >> 6) * (unsigned long long)(kPackedWords * 64), 0, (int)(2u * kTileBytes), 0x00020000);
        // where joint slot x starts inside the descriptor, for all slots at once in lanes 0..6 (as scalar code this is the
        // larger part of the loader's instructions, and the scalar unit is what a CU full of these blocks runs out of)
        const unsigned in_tile = (unsigned)(l0 & 63ull) + ((real && lane < nj) ? (unsigned)lane : 0u);
        const unsigned slot_base = (in_tile >> 6) * kTileBytes + (in_tile & 63u) * 16u;
        const unsigned lane_off = (unsigned)lane * 1024u;
        const bool wanted = lane < (whole_tables ? kPackedWords / 2 : kCappedPairs);
#pragma unroll
        for (int x = 0; x < kTabJointGroup; ++x) {
            const unsigned base = (unsigned)__builtin_amdgcn_readlane((int)slot_base, x);
            if (wanted)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_tab, (lds_ptr)(reinterpret_cast<char*>(&B.jt[x]) + kPackedByte), 16, base + lane_off, 0, 0, 0);
        }
    };
    // header of an item whose loads are in: what the streaming waves read
    auto publish_header = [&](TabBuffer& B, unsigned long long item) __attribute__((always_inline)) {
        const int lane = fresh_lane();
        if (item >= total) {
            if (lane == 0) lds_poke32(lds_offset(&B.hdr.done), 1);
            return;
        }
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        int slen = 0;
        unsigned long long rel = 0ull;
        if (local >= 0) {
            // everything the header needs from what the loads brought, in one LDS round trip: lanes 0..6 read the joints' run
            // counts, lane 7 the trajectory length, lane 8 the plan's row offset (all written by the table pass; packed word 0 =
            // nseg | len, the row offset is the upper half of packed word 11)
            const unsigned pk0 = lds_offset(&B.jt[0]) + kPackedByte;
            const unsigned peek_at = lane < kTabJointGroup ? lds_offset(&B.jt[lane]) + kPackedByte
                                     : (lane == kTabJointGroup ? pk0 + 4u : pk0 + (unsigned)(1 + (kMaxSegments + 1) / 2) * 8u + 4u * ((kMaxSegments + 1) & 1));
            const int peeked = lds_peek32(peek_at);
            const int len = __builtin_amdgcn_readlane(peeked, kTabJointGroup);
            slen = stored_len(len, rows);
            rel = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(peeked, kTabJointGroup + 1) * (unsigned long long)kRowAlign;
            const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
            if (slen > 0 && rel + 4ull * dof * stride > capacity) {
                if (lane == 0 && j0 == 0) atomicOr(&rec.status[first + local], kStatusOverflow);
                slen = 0;
            }
            // the largest run count among the item's joints (wave-uniform): how many passes the expansion needs
            const int runs_of_mine = lane < nj ? peeked : 0;
            int max_runs = 0;
#pragma unroll
            for (int x = 0; x < kTabJointGroup; ++x) {
                const int n_x = __builtin_amdgcn_readlane(runs_of_mine, x);
                max_runs = n_x > max_runs ? n_x : max_runs;
            }
            if (slen > 0 && !whole_tables && max_runs > kCappedRuns) {
                // a capped row whose joint has more than 8 runs inside the cap (short trajectories): fetch the rest now
                if (stamps && lane == 0) stamps[8 * item + 2] = 1ull;
                const unsigned long long l0 = (unsigned long long)local * dof + j0;
                const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<unsigned long long*>(tables) + (l0 >> 6) * (unsigned long long)(kPackedWords * 64), 0, (int)(2u * kTileBytes), 0x00020000);
#pragma unroll
                for (int x = 0; x < kTabJointGroup; ++x) {
                    const unsigned long long li = l0 + (unsigned)(x < nj ? x : 0);
                    const unsigned base = (unsigned)((li >> 6) - (l0 >> 6)) * kTileBytes + (unsigned)(li & 63ull) * 16u;
                    if (lane >= kCappedPairs && lane < kPackedWords / 2)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_tab, (lds_ptr)(reinterpret_cast<char*>(&B.jt[x]) + kPackedByte), 16, base + (unsigned)lane * 1024u, 0, 0, 0);
                }
                LTP_WAIT_VMCNT(0);                          // rare: drain everything (later waits only get easier)
            } else if (stamps && lane == 0 && slen > 0 && !whole_tables) {
                stamps[8 * item + 2] = 0ull;
            }
            // (expanding in the streaming wave that owns the joint instead — seven waves in parallel — was measured: the sampler
            // of 64-sample rows went from 4.1 to 4.9 ms; the streaming waves are the longer side of an item already)
            if (slen > 0) expand_packed_tables(B, nj, max_runs, lane, t_sample);
        }
        if (lane == 0) {
            lds_poke64(lds_offset(&B.hdr.rel), rel);
            lds_poke32(lds_offset(&B.hdr.slen), slen);
            lds_poke32(lds_offset(&B.hdr.j0), j0);
            lds_poke32(lds_offset(&B.hdr.nj), nj);
            lds_poke32(lds_offset(&B.hdr.done), 0);
            lds_poke64(lds_offset(&B.hdr.item), item);
        }
    };
    // waits until every streaming wave is past the item that last used buffer seq % kTabBuffers (lane w < 7 watches wave w)
    auto wait_buffer_free = [&](int seq) __attribute__((always_inline)) {
        if (seq < kTabBuffers) return;
        const int need = seq - kTabBuffers + 1;
        const int lane = fresh_lane();
        const unsigned watch = lds_offset(&s_consumed[lane < kTabStreamWaves ? lane : 0]);
        for (;;) {
            const int c = lds_peek32(watch);
            if (!__builtin_amdgcn_ballot_w64(c < need)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // items in flight: seq (to be published next) .. seq + kTabAhead - 1; their queue positions
    unsigned long long ids[kTabAhead + 1];
#pragma unroll
    for (int d = 0; d < kTabAhead; ++d) {
        ids[d] = next_item_id();
        request(buf[d], ids[d]);
        finish_draw();
    }
    unsigned long long t_prev_pub = 0ull;
    for (int seq = 0;; ++seq) {
        // request item seq + kTabAhead into its buffer, as soon as the streaming waves have left it (the spin loop comes
        // before the queue draw: between the draw's atomic and its use there must be no loop, see next_item_id)
        wait_buffer_free(seq + kTabAhead);
        const unsigned long long t_top = stamps ? wall_clock64() : 0ull;
        ids[kTabAhead] = next_item_id();
        request(buf[(seq + kTabAhead) % kTabBuffers], ids[kTabAhead]);
        const unsigned long long t_req = stamps ? wall_clock64() : 0ull;
        // item seq is in when at most the loads of the kTabAhead younger items are outstanding
        LTP_WAIT_VMCNT(kTabAhead * kTabJointGroup);
        TabBuffer& B = buf[seq % kTabBuffers];
        const unsigned long long item = ids[0];
        if (stamps && fresh_lane() == 0 && item < total) stamps[8 * item + 1] = wall_clock64();
        publish_header(B, item);
        // Publishing is a plain LDS write: the table data is in LDS (the wait above), the header was written by this same
        // lane and LDS serves a wave's requests in order.
        if (fresh_lane() == 0) lds_poke32(lds_offset(&s_ready[seq % kTabBuffers]), seq + 1);
        if (stamps && fresh_lane() == 0 && item < total) { stamps[8 * item + 3] = wall_clock64(); stamps[8 * item + 0] = t_top; stamps[8 * item + 6] = t_req; stamps[8 * item + 7] = t_prev_pub; }
        if (stamps) t_prev_pub = wall_clock64();
        finish_draw();
        if (item >= total) break;                       // the item just published says done: everyone leaves on reading it
#pragma unroll
        for (int d = 0; d < kTabAhead; ++d) ids[d] = ids[d + 1];
    }
    LTP_WAIT_VMCNT(0);                                  // nothing may still be landing in LDS when the wave ends
}

// The register budget decides how many streaming waves a CU holds, and attributes cannot depend on template parameters:
// one kernel per row type. float64: 3 blocks of 8 waves per CU (21 streaming waves, 6 waves per SIMD, <= 80 VGPRs);
// float32 (4 samples per lane in flight, 116 VGPRs; at 80 it spills inside the store loop): 2 blocks (14 streaming waves).
#define LTP_TAB_KERNEL(NAME, ST, TY, WAVES)                                                                                          \
    __global__ void __launch_bounds__(kTabThreads) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))                                    \
    NAME(long long first, long long count, long long base_first, int dof, Records rec, const unsigned long long* __restrict__ offsets, \
         TY* __restrict__ out, unsigned long long capacity, int spread, RowSpec rows, unsigned long long* __restrict__ next_item,     \
         const unsigned long long* __restrict__ tables, int draw_chunk, unsigned long long* __restrict__ stamps, double t_sample)     \
    {                                                                                                                                 \
        sample_tab_body<ST, TY>(first, count, base_first, dof, rec, offsets, out, capacity, spread, rows, next_item, tables, draw_chunk, stamps, t_sample); \
    }
LTP_TAB_KERNEL(k_sample_tab_f64, false, double, 6)
LTP_TAB_KERNEL(k_sample_tab_f64_nt, true, double, 6)
LTP_TAB_KERNEL(k_sample_tab_f32, false, float, 4)
LTP_TAB_KERNEL(k_sample_tab_f32_nt, true, float, 4)
#undef LTP_TAB_KERNEL

// ---------------------------------------------------------------------------------------
// On-device consumer: position envelopes instead of dense rows. A caller that only needs to
// know where each joint can be during each time window of the plan (reachability / limit / collision checks of a
// safety shield, reference README.md:10-13) gets, per plan and joint, [min q, max q] over the samples of each of
// n_windows windows of `window` samples — 16 bytes per window instead of 32 bytes per sample, so nothing the size
// of the dense trajectories ever exists. The values are the minimum and maximum of exactly the q samples k_sample
// would have stored (same run tables, same run_eval expression). Windows that start after the end of the trajectory
// hold its last position (the joint rests there); plans without a trajectory (traj_len 0) get NaN.
// Item = plan x joint group as in k_sample; lane -> (joint, window) task, each walking its samples in order.
// ---------------------------------------------------------------------------------------
LTP_DEV double run_eval_q(const double* c, int m)
{
    const double md = (double)m;
    return __builtin_fma(__builtin_fma(__builtin_fma(c[3], md, c[2]), md, c[1]), md, c[0]);   // the q line of run_eval
}

template <bool PROBE, bool TABLES>
__global__ void __launch_bounds__(kSampleThreads, kSampleBlocksPerCU)
k_envelope(long long first, long long count, long long base_first, int dof, double t_sample, Limits lim, Queries in, Records rec, int window,
           int n_windows, int lg, double* __restrict__ env, unsigned long long* __restrict__ next_item,
           unsigned long long* __restrict__ probe_buf /* diagnostic, PROBE only: 16 stamps per item */,
           const unsigned long long* __restrict__ tables)
{
    __shared__ SegTable tab;
    __shared__ unsigned long long s_item;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    const unsigned long long total = (unsigned long long)count * ngroups;
    constexpr int kChunk = 4;                      // items per queue draw (see k_sample)
    unsigned long long chunk_base = 0ull;
    int chunk_i = 0;
    for (;;) {
        __syncthreads();
        unsigned long long t_top = 0ull;
        if constexpr (PROBE) t_top = wall_clock64();
        if (threadIdx.x == 0) {
            if (chunk_i == 0) chunk_base = atomicAdd(next_item, (unsigned long long)kChunk);
            s_item = chunk_base + (unsigned long long)chunk_i;
        }
        chunk_i = (chunk_i + 1) & (kChunk - 1);
        __syncthreads();
        const unsigned long long item = s_item;
        if (item >= total) break;
        unsigned long long* probe = nullptr;
        if constexpr (PROBE) {
            probe = probe_buf + item * 16;
            if (threadIdx.x == 0) { probe[0] = t_top; probe[1] = wall_clock64(); }
        }
        const int group = (int)(item % ngroups);
        const long long local = (long long)(item / ngroups);
        const long long p = first + local;
        const int j0 = group * kSampleJointGroup;
        const int nj = (dof - j0) < kSampleJointGroup ? (dof - j0) : kSampleJointGroup;
        const int len = rec.traj_len[p];
        const int tasks = nj * n_windows;
        double2_t* const dst = reinterpret_cast<double2_t*>(env) + ((unsigned long long)(p - base_first) * dof + j0) * n_windows;
        if (len <= 0) {
            const double nan = __builtin_nan("");
            for (int task = threadIdx.x; task < tasks; task += kSampleThreads) dst[task] = double2_t{nan, nan};
            continue;
        }
        if constexpr (PROBE) { if (threadIdx.x == 0) probe[2] = wall_clock64(); }
        const ItemRegs<TABLES> regs = fetch_item<TABLES>(p, j0, nj, dof, lim, in, rec, nullptr, tables, first);
        if constexpr (TABLES) install_run_tables(tab, nj, regs.w, t_sample);
        else {
            build_run_tables<PROBE>(tab, p, j0, nj, len, t_sample, lim, rec, regs.pa, regs.pb, probe);
            __syncthreads();
        }
        if constexpr (PROBE) { if (threadIdx.x == 0) probe[9] = wall_clock64(); }
        // g lanes share one (joint, window) task (g = 2^lg divides 64, chosen by the host so that the block has
        // work for all its lanes); lane r of the task takes samples b + r, b + r + g, ... and the g partial results
        // meet in a butterfly. Minimum and maximum do not depend on the order, so any g gives the same bits.
        const int g = 1 << lg;
        for (int base = 0; base < tasks * g; base += kSampleThreads) {
            const int idx = base + (int)threadIdx.x;
            const int task = idx >> lg, r = idx & (g - 1);
            const bool live = task < tasks;
            double lo = __builtin_huge_val(), hi = -__builtin_huge_val();
            if (live) {
                const int jl = task / n_windows, w = task - jl * n_windows;
                const int* st = tab.jt[jl].start;
                const int nruns = tab.jt[jl].nseg;
                const long long b = (long long)w * window;
                const bool past = b >= (long long)len;                            // past the end: the last sample only
                int i = past ? len - 1 + r : (int)b + r;
                const int e = (b + window < (long long)len) ? (int)(b + window) : len;
                int kr = 0, cur = 0, nxt = nruns > 1 ? st[1] : 0x7fffffff;
                // the four q coefficients of the current run stay in registers; they are re-read at a run boundary only
                double c4[4] = {tab.jt[jl].c[0][0], tab.jt[jl].c[0][1], tab.jt[jl].c[0][2], tab.jt[jl].c[0][3]};
                for (; i < e; i += g) {
                    if (nxt <= i) {
                        do {
                            ++kr;
                            cur = nxt;
                            nxt = kr + 1 < nruns ? st[kr + 1] : 0x7fffffff;
                        } while (nxt <= i);
#pragma unroll
                        for (int x = 0; x < 4; ++x) c4[x] = tab.jt[jl].c[kr][x];
                    }
                    const double q = run_eval_q(c4, i - cur + 1);
                    lo = __builtin_fmin(lo, q);
                    hi = __builtin_fmax(hi, q);
                }
            }
            for (int d = 1; d < g; d <<= 1) {
                lo = __builtin_fmin(lo, __shfl_xor(lo, d));
                hi = __builtin_fmax(hi, __shfl_xor(hi, d));
            }
            if (live && r == 0) dst[task] = double2_t{lo, hi};
        }
        if constexpr (PROBE) {
            __syncthreads();
            if (threadIdx.x == 0) probe[10] = wall_clock64();
        }
    }
}

// ---------------------------------------------------------------------------------------
// Receding horizon (reference README.md:10-13): the start state of the next plan is the state
// at sample k of the previous trajectory, gathered on the device without a host round trip.
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256)
k_replan_states(long long first, long long count, int dof, RowSpec rows, Queries in, Records rec,
                const unsigned long long* __restrict__ offsets, const T* __restrict__ tile, unsigned long long capacity,
                const int* __restrict__ sample_index, int uniform_index,
                double* __restrict__ q_0, double* __restrict__ v_0, double* __restrict__ a_0, long long sq, long long sj)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * dof) return;
    const long long local = idx / dof;
    const int j = (int)(idx - local * dof);
    const long long p = first + local;
    const long long dst = local * sq + (long long)j * sj;
    const int slen = stored_len(rec.traj_len[p], rows);
    const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    const unsigned long long rel = offsets[p] - offsets[first];
    // not sampled: no trajectory, flagged by the sampler as not fitting its tile, or (the same test k_sample applies)
    // rows that would end beyond the tile -> carry the start state over unchanged, read nothing outside the tile
    if (slen <= 0 || (rec.status[p] & kStatusOverflow) || rel + 4ull * dof * stride > capacity) {
        const long long ix = p * in.sq + (long long)j * in.sj;
        q_0[dst] = in.q_0[ix];
        v_0[dst] = in.v_0[ix];
        a_0[dst] = in.a_0[ix];
        return;
    }
    int k = sample_index ? sample_index[local] : uniform_index;
    k = k < 0 ? 0 : (k >= slen ? slen - 1 : k);   // beyond the stored samples: the last stored state
    const T* row = tile + rel + (unsigned long long)j * stride + k;
    const unsigned long long arr = (unsigned long long)dof * stride;
    q_0[dst] = (double)row[0];
    v_0[dst] = (double)row[arr];
    a_0[dst] = (double)row[2 * arr];
}

// One lane walks the runs of one joint in order: the same cut points, jerk_at(), run_coef() and run_eval() as the
// cooperative table build of k_sample, with everything in registers. A kernel whose work per plan is small
// (k_state_at) uses this instead of LDS tables: no block-level build, no per-item latency, 64 independent
// (plan, joint) pairs per wave. (For the envelope consumer the same form is slower than the cooperative kernel,
// 26.8 vs 20.0 ms per 1 M plans: lanes of a wave sit in runs of different lengths.) visit(b, e, rc) is called for every run [b, e) with its
// coefficients and returns true to stop; (a, v, q) hold the state before the run and are advanced to its last sample
// (exactly the value the sampler stores there) after each call that returns false.
// MATLAB's mod(x, y) for y > 0 as LTPlanner.m:531 uses it: x - floor(x./y).*y, except that "if y is not an integer and the
// quotient x./y is within roundoff error of an integer, then n is that integer" (MATLAB documentation), i.e. the result is 0;
// the round-off test is GNU Octave's published rule, as in the test suite's CPU twin. (cc:747 has no such rule.)
LTP_DEV double matlab_mod(double x, double y)
{
    if (y == 0.0) return x;
    const double q = x / y;
    const double n = __builtin_rint(q);
    if (__builtin_rint(y) != y && dabs((q - n) / n) < kDblEps) return 0.0;
    return x - y * dfloor(q);
}

template <int SEM = kSemCpp, class Visit>
LTP_DEV void for_each_run(const Limits& lim, const Records& rec, long long rj, int j, int len, double Ts, double& q, double& v,
                          double& a, Visit&& visit, bool last_joint = true)
{
    int sw[7];                                                                        // sampled switch indices (cc:751-757)
    double fr[7], frts[7];
#pragma unroll
    for (int x = 0; x < 7; ++x) {
        const double tk = rec.t_scaled[rj * 7 + x];
        fr[x] = SEM == kSemMatlab ? matlab_mod(tk, Ts) : tk - Ts * dfloor(tk / Ts);   // cc:747 / LTPlanner.m:531
        frts[x] = fr[x] / Ts;
        sw[x] = (x & 1) ? (int)dceil(tk / Ts) : (int)dfloor(tk / Ts);
    }
    const double dir = rec.dir[rj];
    const double dj = dir * lim.j_max[j];
    const double vsnap = rec.v_drive[rj] * dir;                                       // cc:823
    const bool modp = (double)rec.mod[rj] == 1.0;
    // phase jerks (cc:735-744) and the nine possible correction terms (cc:771-807), as in build_run_tables step (2)
    const double J0 = dj * (modp ? -1.0 : 1.0), J2 = dj * (modp ? 1.0 : -1.0), J4 = dj * -1.0, J6 = dj * 1.0;
    const double Jp[7] = {J0, dj * 0.0, J2, dj * 0.0, J4, dj * 0.0, J6};
    const double d20 = (fr[2] - fr[0]) / Ts;
    const double corr[9] = {frts[0] * J0, (1 - frts[1]) * J2, frts[2] * J2, d20 * J2, (1 - frts[3]) * J4,
                            frts[4] * J4, 0.0, (1 - frts[5]) * J6, frts[6] * J6};
    // Candidate cut points: every index where the jerk array or a snap rule (cc:815-829) can change — the same set as kCutBase /
    // kCutDelta of the cooperative build: per sampled switch index s_g a few CONSECUTIVE integers s_g + lo_g .. s_g + hi_g. The next
    // cut after b inside group g is therefore max(s_g + lo_g, b + 1) if that is <= s_g + hi_g: three operations per group instead of
    // four per candidate. MATLAB semantics: the corrections sit one sample earlier, the constant-velocity samples are s2 .. s3-2
    // and the tail starts at s6 (LTPlanner.m:616, 620).
    constexpr int cut_lo[7] = {0, 0, 0, -1, 0, 0, 0}, cut_hi[7] = {2, 1, 2, 1, 2, 1, 2};
    constexpr int mcut_lo[7] = {0, -1, 0, -1, -1, -1, 0}, mcut_hi[7] = {1, 0, 1, 0, 1, 0, 1};
    int glo[7], ghi[7];
#pragma unroll
    for (int g = 0; g < 7; ++g) {
        glo[g] = sw[g] + (SEM == kSemMatlab ? mcut_lo[g] : cut_lo[g]);
        ghi[g] = sw[g] + (SEM == kSemMatlab ? mcut_hi[g] : cut_hi[g]);
    }
    const bool phase4 = sw[3] - sw[2] > 2;                                            // cc:813
    int b = 0;
    for (int run = 0; run < kMaxSegments && b < len; ++run) {
        int e = len;                                                                  // next cut point after b
#pragma unroll
        for (int g = 0; g < 7; ++g) {
            const int c = glo[g] > b ? glo[g] : b + 1;
            e = (c <= ghi[g] && c < e) ? c : e;
        }
        int mode = 0;
        if constexpr (SEM == kSemMatlab) {
            if (b >= sw[6]) mode |= last_joint ? kModeTail : (kModeTail | kModeKeepA);
            if (phase4 && b >= sw[2] && b < sw[3] - 1) mode |= kModeVSnap;
        } else {
            if (b > sw[6]) mode |= kModeTail;
            if (phase4 && b >= sw[2] + 1 && b < sw[3] - 1) mode |= kModeVSnap;
        }
        const double J = jerk_at<SEM>(sw, Jp, corr, b);
        const RunCoef rc = run_coef<SEM>(mode, J, a, v, q, vsnap, Ts);
        if (visit(b, e, rc)) return;
        if constexpr (SEM == kSemMatlab) {
            // LTPlanner.m:604-624: a, v, q are cumulative sums over the arrays as they stand — behind the constant-velocity
            // samples (and in the tail) v continues from the UN-snapped sum, the acceleration sum never stops
            const double md = (double)(e - b);
            const double v2 = 0.5 * (Ts * (Ts * J));
            const double v_cum = __builtin_fma(__builtin_fma(v2, md, Ts * a + v2), md, v);   // run_eval's v of an ordinary run
            const double a_cum = __builtin_fma(Ts * J, md, a);
            double vv, aa, jj;
            run_eval(rc.c, e - b, q, vv, aa, jj);
            v = v_cum;
            a = a_cum;
        } else {
            double jj;
            run_eval(rc.c, e - b, q, v, a, jj);
        }
        b = e;
    }
}

// Receding horizon without any sampled rows: the state (q, v, a) at trajectory sample k of every plan straight from
// the switching-time records. A caller that only needs the restart state pays neither the table build of a sampler
// item (~15 us of latency per plan) nor a byte of trajectory traffic. The result has the bits of the row element the
// sampler would have stored at k.
template <int SEM>
__global__ void __launch_bounds__(256)
k_state_at(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
           const int* __restrict__ sample_index, int uniform_index,
           double* __restrict__ q_0, double* __restrict__ v_0, double* __restrict__ a_0, long long sq, long long sj)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * dof) return;
    const long long local = idx / dof;
    const int j = (int)(idx - local * dof);
    const long long p = first + local;
    const long long dst = local * sq + (long long)j * sj;
    const long long ix = p * in.sq + (long long)j * in.sj;
    double q = in.q_0[ix], v = in.v_0[ix], a = in.a_0[ix];   // state "before sample 0" (cc:810-812)
    const int len = rec.traj_len[p];
    if (len > 0) {
        int k = sample_index ? sample_index[local] : uniform_index;
        k = k < 0 ? 0 : (k >= len ? len - 1 : k);             // beyond the end: the last state
        for_each_run<SEM>(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [&](int b, int e, const RunCoef& rc) {
            if (k >= e) return false;
            double jj;
            run_eval(rc.c, k + 1 - b, q, v, a, jj);
            return true;
        }, j == dof - 1);
    }
    q_0[dst] = q;
    v_0[dst] = v;
    a_0[dst] = a;
}

// planTrajectory's end-limit check (cc:59-61) without sampled rows: lane = (plan, joint) walks its runs to the last
// trajectory sample — the bits k_sample would have stored at traj_len-1, which is also what build_run_tables step (5)
// tests — and flags the plan if that position lies outside the joint range.
__global__ void __launch_bounds__(256)
k_end_limit(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * dof) return;
    const long long local = idx / dof;
    const int j = (int)(idx - local * dof);
    const long long p = first + local;
    const int len = rec.traj_len[p];
    if (len <= 0) return;                                     // failed before sampling: the reference never gets to cc:59
    const long long ix = p * in.sq + (long long)j * in.sj;
    double q = in.q_0[ix], v = in.v_0[ix], a = in.a_0[ix];
    for_each_run(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [](int, int, const RunCoef&) { return false; });
    if (q < lim.q_min[j] || q > lim.q_max[j]) atomicOr(&rec.status[p], kStatusEndLimit);
}

// The table pass: the run tables of plans [first, first + count) as a kernel of its own, lane = (plan, joint), everything
// in registers (the walk of k_state_at), written word for word in the JointTable layout. A sampler item then costs one
// (prefetched) table read instead of a cooperative build of ~8 us of latency — what short rows, the envelope consumer and
// receding-horizon rows are bound by. 912 bytes per joint (packed): worth it when a plan's rows are not much longer than that.
// Also applies the end-limit check of cc:59-61 (the sampler variants that read tables no longer do).
template <int SEM>
__global__ void __launch_bounds__(256)
k_build_tables(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
               int needed_end /* runs that start at or after this sample are not stored (capped rows) */,
               const unsigned long long* __restrict__ offsets /* nullptr: no row offsets wanted */, long long base_first,
               unsigned long long* __restrict__ tables)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * dof) return;
    const long long local = idx / dof;
    const int j = (int)(idx - local * dof);
    const long long p = first + local;
    const unsigned long long lane_id = (unsigned long long)idx;
    auto word = [&](int w) -> unsigned long long* { return tables + table_word_index(lane_id, w); };
    const int len = rec.traj_len[p];
    typedef double pair_t __attribute__((ext_vector_type(2)));
    auto store_pair = [&](int w, double lo, double hi) {        // 16 bytes per lane: a full 1 KiB line per wave instruction
        pair_t v2;
        v2[0] = lo;
        v2[1] = hi;
        __builtin_nontemporal_store(v2, reinterpret_cast<pair_t*>(word(w)));
    };
    if (len <= 0) { *word(0) = 0ull; return; }                 // nseg 0: the sampler skips such plans anyway
    const long long ix = p * in.sq + (long long)j * in.sj;
    double q = in.q_0[ix], v = in.v_0[ix], a = in.a_0[ix];
    store_pair(12, rec.v_drive[p * dof + j] * rec.dir[p * dof + j], 0.0);   // vsnap, as for_each_run forms it (cc:823)
    // Packed runs: five words each, stored as word pairs two runs at a time. A lane whose runs are past the cap stores zeros as
    // long as a neighbour still stores: the lanes of a wave are the lanes of one table tile, and a 1 KiB line written whole costs
    // HBM half of what the same line written by some of its lanes does (measured: 1.53 -> 1.1 ms for the same tables).
    int run = 0, slots = 0;
    int last_b = len;
    double ha = 0.0, hv = 0.0, hq = 0.0, hj = 0.0, hm = 0.0;     // the even run of a pair, until its odd partner arrives
    auto as_word = [](int mode) { return __builtin_bit_cast(double, (unsigned long long)(unsigned)mode); };
    for_each_run<SEM>(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [&](int b, int, const RunCoef& rc) {
        const bool mine = b < needed_end;
        if (__builtin_amdgcn_ballot_w64(mine) != 0ull) {
            // q, v, a still hold the state before this run: for_each_run advances them after the visit
            const double sa = mine ? a : 0.0, sv = mine ? v : 0.0, sq = mine ? q : 0.0, sj = mine ? rc.c[9] : 0.0;
            const double sm = mine ? as_word(rc.mode) : 0.0;
            if (mine) {
                reinterpret_cast<int*>(word(1 + (run >> 1)))[run & 1] = b;
                ++run;
            }
            if (slots & 1) {
                const int w0 = kPackedHeaderWords + (slots - 1) * kPackedRunWords;   // even: a pair boundary
                store_pair(w0, ha, hv); store_pair(w0 + 2, hq, hj); store_pair(w0 + 4, hm, sa); store_pair(w0 + 6, sv, sq); store_pair(w0 + 8, sj, sm);
            } else {
                ha = sa; hv = sv; hq = sq; hj = sj; hm = sm;
            }
            ++slots;
        }
        if (!mine && last_b == len) last_b = b;                  // first run that is not stored: it ends the last stored one
        return false;                                          // the walk still goes to the last sample: end-limit check
    }, j == dof - 1);
    if (slots & 1) {
        const int w0 = kPackedHeaderWords + (slots - 1) * kPackedRunWords;
        store_pair(w0, ha, hv); store_pair(w0 + 2, hq, hj); store_pair(w0 + 4, hm, 0.0);
    }
    static_assert(kPackedHeaderWords % 2 == 0 && (2 * kPackedRunWords) % 2 == 0, "two runs start on a word pair");
    reinterpret_cast<int*>(word(1 + (run >> 1)))[run & 1] = last_b;
    *word(0) = (unsigned long long)(unsigned)run | ((unsigned long long)(unsigned)len << 32);
    // where the plan's rows start inside the range the sampler is called for: what k_sample_tab's loader would otherwise
    // have to load per item (plan sizes are multiples of kRowAlign elements)
    // (saturated: an offset that does not fit 32 bits is beyond any tile, and the sampler then flags the plan as not fitting)
    if (offsets) {
        const unsigned long long rel = (offsets[p] - offsets[base_first]) / kRowAlign;
        reinterpret_cast<unsigned*>(word(1 + (kMaxSegments + 1) / 2))[(kMaxSegments + 1) & 1] = rel > 0xffffffffull ? 0xffffffffu : (unsigned)rel;
    }
    if constexpr (SEM == kSemCpp) {                            // LTPlanner.m has no position limits, hence no end-limit check
        if (q < lim.q_min[j] || q > lim.q_max[j]) atomicOr(&rec.status[p], kStatusEndLimit);   // cc:59-61: q is sample len-1
    }
}

// ---------------------------------------------------------------------------------------
// The whole of planTrajectory (cc:7-63) for a handful of queries in ONE launch of ONE block — what a single
// LongTermPlanner::planTrajectory call is (BASELINE.json configs[0]). The batched path spends nine launches, two stream
// synchronisations and two PCIe copies on such a call (140 us against ~36 us on one CPU core); here the inputs are read
// from, and records and rows are written straight into, host memory that the device can address (pinned), and the host
// waits for one word. Same device functions as the batched kernels, so the results have the same bits:
//   lane = (query, joint): checkInputs + optSwitchTimes with the root finder          cc:14-30
//   lane = query: slowest joint                                                        cc:31-39
//   lane = (query, joint): timeScaling, all candidates in the reference's order       cc:43-55, 358-645
//   one lane: trajectory lengths, packed offsets                                       cc:716-719
//   per plan and joint: build_run_tables + stream_rows                                 cc:59-61, 706-841
// n * dof <= kSmallPairs. With rows the grid has one block per joint (at most kSmallBlocks): every block repeats the cheap
// stages above for itself (no communication) and then builds the tables and writes the rows of its own joints only — one
// block moves ~25 GB/s into host memory, a plan's 386 KB would take it as long as everything else together. The end-limit
// bits each block finds go to io.end_flags (the host ORs them into status); the last block to arrive reports completion.
// ---------------------------------------------------------------------------------------
constexpr int kSmallPairs = 128;
constexpr int kSmallBlocks = 8;
struct SmallHost {               // device-addressable host memory (or device memory), all caller-owned
    const double* in[4];         // q_goal, q_0, v_0, a_0: row-major [n][dof]
    Records rec;                 // [n][dof][7] ... as in the batched ABI
    unsigned long long* offsets; // [n + 1]
    double* rows;                // packed trajectories, `capacity` doubles; nullptr: do not sample (end-limit check only)
    unsigned long long capacity;
    int* end_flags;              // [gridDim.x][n]: LTP_STATUS_END_LIMIT bits found by each block for the joints it sampled
    unsigned int* arrivals;      // device memory, zero between launches: blocks that have finished
    volatile int* done;          // set to 1 (2: rows did not fit `capacity`, nothing sampled) when everything above is visible
};

struct SmallShared {             // LDS of one small-batch block
    SegTable tab;
    double t_opt[kSmallPairs][7], t_scaled[kSmallPairs][7], dir[kSmallPairs], vd[kSmallPairs];
    double treq[kSmallPairs];
    signed char mod[kSmallPairs];
    int flags[kSmallPairs], slowest[kSmallPairs], len[kSmallPairs], status[kSmallPairs];
    unsigned long long off[kSmallPairs + 1];
    int fit;
    unsigned long long tick[8];     // diagnostic: wall clock of thread 0 at the phase boundaries
};

// GIVEN: the switching-time records are inputs (LongTermPlanner::getTrajectory, cc:706-841: t_scaled, dir, mod, v_drive
// from io.rec, start states from io.in[1..3]); stages 1-3 are skipped, lengths are those of k_finalize.
// Every thread of every block of the grid calls this (it contains block barriers); on return the block's part is done and,
// in the last block to finish, *io.done has been set.
template <bool GIVEN>
LTP_DEV void plan_small_body(int n, int dof, double t_sample, int goal_check, RowSpec rows, const Limits& lim, const SmallHost& io,
                             SmallShared& sh)
{
    SegTable& tab = sh.tab;
    double (&s_t_opt)[kSmallPairs][7] = sh.t_opt;
    double (&s_t_scaled)[kSmallPairs][7] = sh.t_scaled;
    double (&s_dir)[kSmallPairs] = sh.dir;
    double (&s_vd)[kSmallPairs] = sh.vd;
    double (&s_treq)[kSmallPairs] = sh.treq;
    signed char (&s_mod)[kSmallPairs] = sh.mod;
    int (&s_flags)[kSmallPairs] = sh.flags;
    int (&s_slowest)[kSmallPairs] = sh.slowest;
    int (&s_len)[kSmallPairs] = sh.len;
    int (&s_status)[kSmallPairs] = sh.status;
    unsigned long long (&s_off)[kSmallPairs + 1] = sh.off;
    int& s_fit = sh.fit;
    const int t = threadIdx.x;
    const int pairs = n * dof;
    // (query, joint) pair `pid` of stages 1-3: lane pid. Spreading the pairs of a single call over the four waves of the block
    // (so that joints in different branches of optSwitchTimes / timeScaling run side by side) was measured and is SLOWER by 1.8x:
    // the kernel is ~530 KB of straight-line code behind a 64 KB instruction cache, and four waves in four places of it wait for
    // instruction fetches more than one wave walking through it (single call)
    const int pid = t;
    const bool pair = pid < pairs;
    const int q = pair ? pid / dof : 0, j = pair ? pid - q * dof : 0;
    JointLimits L = {0.0, 0.0, 0.0, 0.0, 0.0};
    double qg = 0.0, q0 = 0.0, v0 = 0.0, a0 = 0.0;
    if constexpr (GIVEN) {
        if (t < n) { s_status[t] = 0; s_len[t] = 0; s_treq[t] = 0.0; s_slowest[t] = -1; }
        __syncthreads();
        if (pair) {
            L = load_limits(lim, j);
            q0 = io.in[1][pid]; v0 = io.in[2][pid]; a0 = io.in[3][pid];
            double ts[7];
#pragma unroll
            for (int k = 0; k < 7; ++k) { ts[k] = io.rec.t_scaled[pid * 7 + k]; s_t_scaled[pid][k] = ts[k]; s_t_opt[pid][k] = 0.0; }
            s_dir[pid] = io.rec.dir[pid];
            s_vd[pid] = io.rec.v_drive[pid];
            s_mod[pid] = io.rec.mod[pid];
            const int l = joint_len(ts, t_sample);
            if (l < 0) atomicOr(&s_status[q], kStatusNonFinite);
            else atomicMax(&s_len[q], l);
        }
        __syncthreads();
    } else {
    if (t == 0) sh.tick[0] = (unsigned long long)wall_clock64();
    // ---- stage 1 ----
    if (pair) {
        L = load_limits(lim, j);
        qg = io.in[0][pid]; q0 = io.in[1][pid]; v0 = io.in[2][pid]; a0 = io.in[3][pid];
        int flags = check_inputs_joint(L, q0, v0, a0) ? 0 : kStatusInvalidInput;
        if (goal_check && !(qg >= L.q_min && qg <= L.q_max)) flags |= kStatusGoalOutside;
        double tt[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double dir = 0.0;
        int mod = 0;
        MatlabCtx mc;
        if (opt_switch_times<true>(L.a_max, L.j_max, L.v_max, t_sample, qg, q0, v0, a0, L.v_max, tt, dir, mod, mc) == kOptFalse) flags |= kStatusOptFailed;
#pragma unroll
        for (int k = 0; k < 7; ++k) s_t_opt[pid][k] = tt[k];
        s_dir[pid] = dir;
        s_mod[pid] = (signed char)mod;
        s_flags[pid] = flags;
    }
    __syncthreads();
    if (t == 0) sh.tick[1] = (unsigned long long)wall_clock64();
    // ---- slowest joint (cc:31-39: strict '>', first index wins, NaN never wins, init -1) ----
    if (t < n) {
        double best_t = -1.0;
        int best_j = -1, flags = 0;
        for (int jj = 0; jj < dof; ++jj) {
            const double t6 = s_t_opt[t * dof + jj][6];
            flags |= s_flags[t * dof + jj];
            if (t6 > best_t) { best_t = t6; best_j = jj; }
        }
        if (best_j < 0) flags |= kStatusNoSlowest;
        s_treq[t] = best_t;
        s_slowest[t] = best_j;
        s_status[t] = flags;
        s_len[t] = 0;
    }
    __syncthreads();
    // ---- time scaling + fallback (cc:43-55) ----
    if (pair) {
        const int flags = s_status[q];
        double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double vd = L.v_max;
        int mod = 0;                                   // failed query: zero record, never sampled
        if (flags == 0) {
            mod = s_mod[pid];
            if (j != s_slowest[q]) {
                int which = 0;
                MatlabCtx mc;
                time_scaling_full(L, t_sample, qg, q0, v0, a0, s_dir[pid], s_treq[q], vd, ts, mod, which, mc);
            }
            double mx = ts[0];
#pragma unroll
            for (int k = 1; k < 7; ++k) if (mx < ts[k]) mx = ts[k];
            if (mx <= 0.0) {
#pragma unroll
                for (int k = 0; k < 7; ++k) ts[k] = s_t_opt[pid][k];
            }
            const int l = joint_len(ts, t_sample);
            if (l < 0) atomicOr(&s_status[q], kStatusNonFinite);
            else atomicMax(&s_len[q], l);
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) s_t_scaled[pid][k] = ts[k];
        s_vd[pid] = vd;
        s_mod[pid] = (signed char)mod;
    }
    __syncthreads();
    if (t == 0) sh.tick[2] = (unsigned long long)wall_clock64();
    }   // !GIVEN
    // ---- lengths and packed offsets ----
    if (t == 0) {
        unsigned long long run = 0ull;
        for (int p = 0; p < n; ++p) {
            if (s_status[p] != 0) s_len[p] = 0;
            s_off[p] = run;
            run += plan_size(stored_len(s_len[p], rows), dof);
        }
        s_off[n] = run;
        s_fit = (io.rows == nullptr || run <= io.capacity) ? 1 : 0;
    }
    __syncthreads();
    // records the sampler needs, in the shape of the batched ABI but in LDS (generic pointers)
    Records lrec;
    lrec.t_opt = &s_t_opt[0][0]; lrec.t_scaled = &s_t_scaled[0][0]; lrec.dir = s_dir; lrec.v_drive = s_vd; lrec.mod = s_mod;
    lrec.t_required = s_treq; lrec.slowest = s_slowest; lrec.traj_len = s_len; lrec.status = s_status;
    if (io.rows != nullptr && s_fit) {
        // ---- getTrajectory + end-limit check: this block's joints of every plan ----
        for (int p = 0; p < n; ++p) {
            const int len = s_len[p];
            if (len <= 0) continue;
            const int slen = stored_len(len, rows);
            const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
            double* const base = io.rows + s_off[p];
            const int written = (slen + 1) / 2 * 2;
            const int padn = (int)stride - written;
            for (int jj = blockIdx.x; jj < dof; jj += gridDim.x) {
                // what fetch_item<false> would have loaded for this lane (one joint: joint slot 0 = threads 0..31)
                const int k = t & 31;
                double pa = 0.0, pb = 0.0;
                if (t < 32) {
                    const int rj = p * dof + jj;
                    if (k < 7) pa = s_t_scaled[rj][k];
                    else if (k == 7) { pa = s_dir[rj]; pb = lim.j_max[jj]; }
                    else if (k == 8) { pa = s_vd[rj]; pb = s_dir[rj]; }
                    else if (k == 9) pa = io.in[1][rj];
                    else if (k == 10) pa = io.in[2][rj];
                    else if (k == 11) pa = io.in[3][rj];
                    else if (k == 12) pa = (double)s_mod[rj];
                }
                build_run_tables(tab, p, jj, 1, len, t_sample, lim, lrec, pa, pb);
                __syncthreads();
                stream_rows<false, false, double>(tab, jj, 1, dof, slen, stride, base, rows);
                // row padding beyond the last 16-byte slot: zero, so that the packed buffer is deterministic
                for (int e = t; e < 4 * padn; e += kSampleThreads)
                    base[((unsigned long long)(e / padn) * dof + jj) * stride + written + e % padn] = 0.0;
                __syncthreads();
            }
        }
        if (t < n) io.end_flags[blockIdx.x * n + t] = s_status[t] & kStatusEndLimit;
    } else if (pair && s_len[q] > 0) {
        // no rows wanted: the end-limit check alone (k_end_limit). (For a single call the cooperative table build — 32 lanes per
        // joint, the verdict from its step (5) — was measured in this place: 9.9 us against 10.7 us for this walk; not kept.)
        double qq = q0, vv = v0, aa = a0;
        for_each_run(lim, lrec, pid, j, s_len[q], t_sample, qq, vv, aa, [](int, int, const RunCoef&) { return false; });
        if (qq < L.q_min || qq > L.q_max) atomicOr(&s_status[q], kStatusEndLimit);
    }
    __syncthreads();
    if (t == 0) sh.tick[3] = (unsigned long long)wall_clock64();
    // ---- records out (block 0; the other blocks computed the same values) ----
    if (blockIdx.x == 0) {
    if constexpr (!GIVEN) {
        if (t < pairs) {
#pragma unroll
            for (int k = 0; k < 7; ++k) { io.rec.t_opt[t * 7 + k] = s_t_opt[t][k]; io.rec.t_scaled[t * 7 + k] = s_t_scaled[t][k]; }
            io.rec.dir[t] = s_dir[t];
            io.rec.v_drive[t] = s_vd[t];
            io.rec.mod[t] = s_mod[t];
        }
        if (t < n) {
            io.rec.t_required[t] = s_treq[t];
            io.rec.slowest[t] = s_slowest[t];
        }
    }
    if (t < n) {
        io.rec.traj_len[t] = s_len[t];
        io.rec.status[t] = s_status[t];
    }
    if (t <= n) io.offsets[t] = s_off[t];
    }
    __threadfence_system();
    __syncthreads();
    if (t == 0) sh.tick[4] = (unsigned long long)wall_clock64();
    if (t == 0) {
        // the last block to get here has seen every other block's fence: it reports, and re-arms the counter
        if (atomicAdd(io.arrivals, 1u) == gridDim.x - 1) {
            *io.arrivals = 0u;
            __threadfence_system();
            *io.done = s_fit ? 1 : 2;
            __threadfence_system();
        }
    }
}

template <bool GIVEN>
__global__ void __launch_bounds__(kSampleThreads)
k_plan_small(int n, int dof, double t_sample, int goal_check, RowSpec rows, Limits lim, SmallHost io)
{
    __shared__ SmallShared sh;
    plan_small_body<GIVEN>(n, dof, t_sample, goal_check, rows, lim, io, sh);
}

int small_batch_pairs() { return kSmallPairs; }
int small_batch_blocks(int dof, bool with_rows) { return !with_rows ? 1 : (dof < kSmallBlocks ? (dof > 0 ? dof : 1) : kSmallBlocks); }

void launch_plan_small(hipStream_t s, int n, int dof, double t_sample, int goal_check, RowSpec rows, Limits lim, const double* const in[4],
                       Records rec, unsigned long long* offsets, double* out_rows, unsigned long long capacity, int* end_flags,
                       unsigned int* arrivals, volatile int* done, bool records_given)
{
    SmallHost io;
    for (int k = 0; k < 4; ++k) io.in[k] = in[k];
    io.rec = rec; io.offsets = offsets; io.rows = out_rows; io.capacity = capacity; io.end_flags = end_flags; io.arrivals = arrivals;
    io.done = done;
    const dim3 grid((unsigned)small_batch_blocks(dof, out_rows != nullptr));
    if (records_given) hipLaunchKernelGGL(k_plan_small<true>, grid, dim3(kSampleThreads), 0, s, n, dof, t_sample, goal_check, rows, lim, io);
    else hipLaunchKernelGGL(k_plan_small<false>, grid, dim3(kSampleThreads), 0, s, n, dof, t_sample, goal_check, rows, lim, io);
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
// items per queue draw: 1 for whole trajectories; for capped rows as many as keep a draw at >= ~256 KB of rows, at most 8
static int queue_draw_chunk(RowSpec rows, bool f32, int joints_per_item)
{
    if (rows.max_samples <= 0) return 1;
    const long long item_bytes = 4ll * (f32 ? 4 : 8) * rows.max_samples * joints_per_item;
    int k = 1;
    while (k < 8 && item_bytes * (2 * k) <= 262144) k *= 2;
    return k;
}

// how many blocks of a persistent (work-queue) kernel the device holds at once: 0 = k_sample float64 rows,
// 1 = k_sample float32 rows, 2 = k_envelope
int sample_resident_blocks(int device, int which)
{
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    hipError_t e;
    if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample<true, false, float>, kSampleThreads, 0);
    else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_envelope<false, false>, kSampleThreads, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample<true, false, double>, kSampleThreads, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 4;
    return cus * per_cu;
}

unsigned long long table_bytes(long long lanes)
{
    return (unsigned long long)((lanes + 63) / 64) * (unsigned long long)kPackedWords * 64ull * 8ull;
}

void launch_build_tables(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
                         RowSpec rows, bool whole_trajectory, const unsigned long long* offsets, long long base_first, unsigned long long* tables,
                         int semantics)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    // capped rows only touch the samples before max_samples * stride
    long long needed = 0x7fffffffll;
    if (!whole_trajectory && rows.max_samples > 0) needed = (long long)rows.max_samples * (rows.stride > 1 ? rows.stride : 1);
    if (needed > 0x7fffffffll) needed = 0x7fffffffll;
    if (semantics == kSemMatlab)
        hipLaunchKernelGGL(k_build_tables<kSemMatlab>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, first, count, dof, t_sample, lim, in, rec,
                           (int)needed, offsets, base_first, tables);
    else
        hipLaunchKernelGGL(k_build_tables<kSemCpp>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, first, count, dof, t_sample, lim, in, rec,
                           (int)needed, offsets, base_first, tables);
}

int sample_tab_resident_blocks(int device, bool f32)
{
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    hipError_t e = f32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample_tab_f32_nt, kTabThreads, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample_tab_f64_nt, kTabThreads, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = f32 ? 2 : 3;
    return cus * per_cu;
}

void launch_sample_tab(hipStream_t s, long long first, long long count, long long base_first, int dof, Records rec,
                       const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity, int flags, RowSpec rows,
                       unsigned long long* next_item, int resident_blocks, const unsigned long long* tables, double t_sample,
                       unsigned long long* stamps)
{
    if (count <= 0) return;
    int spread = (flags >> 8) & 0xFFFF;
    if (spread == 0) spread = kSampleSpread;
    if ((long long)spread > count) spread = (int)count;
    const int ngroups = (dof + kTabJointGroup - 1) / kTabJointGroup;
    long long blocks = resident_blocks > 0 ? resident_blocks : 768;
    if (blocks > count * ngroups) blocks = count * ngroups;
    const dim3 grid((unsigned)blocks), block(kTabThreads);
    // (the loader pays one exposed atomic round trip per draw: larger chunks than k_sample's)
    const int draw_chunk = 2 * queue_draw_chunk(rows, f32, dof < kTabJointGroup ? dof : kTabJointGroup);
#define LTP_TAB_CASE(K, TY) hipLaunchKernelGGL(K, grid, block, 0, s, first, count, base_first, dof, rec, offsets, (TY*)out, capacity, spread, rows, next_item, tables, draw_chunk, stamps, t_sample)
    switch ((flags & 1) | (f32 ? 2 : 0)) {
    case 0: LTP_TAB_CASE(k_sample_tab_f64, double); break;
    case 1: LTP_TAB_CASE(k_sample_tab_f64_nt, double); break;
    case 2: LTP_TAB_CASE(k_sample_tab_f32, float); break;
    default: LTP_TAB_CASE(k_sample_tab_f32_nt, float); break;
    }
#undef LTP_TAB_CASE
}

void launch_sample(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in,
                   Records rec, const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity,
                   int flags, RowSpec rows, unsigned long long* next_item, int resident_blocks, unsigned long long* stamps)
{
    if (count <= 0) return;
    int spread = (flags >> 8) & 0xFFFF;
    if (spread == 0) spread = kSampleSpread;
    if ((long long)spread > count) spread = (int)count;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    long long blocks = resident_blocks > 0 ? resident_blocks : 1536;
    if (blocks > count * ngroups) blocks = count * ngroups;
    const dim3 grid((unsigned)blocks);
    const dim3 block(kSampleThreads);
    // flags bit 0: non-temporal stores; bit 1 (diagnostic): skip the arithmetic and store sample indices, which
    // measures the ceiling of this store pattern; bits 8..23: block interleave factor (0 = default 64, 1 = plan order)
    const int draw_chunk = queue_draw_chunk(rows, f32, dof < kSampleJointGroup ? dof : kSampleJointGroup);
#define LTP_SAMPLE_CASE(ST, DR, TY) hipLaunchKernelGGL((k_sample<ST, DR, TY>), grid, block, 0, s, first, count, dof, t_sample, lim, in, rec, offsets, (TY*)out, capacity, stamps, spread, rows, next_item, draw_chunk)
    switch ((flags & 3) | (f32 ? 4 : 0)) {
    case 0: LTP_SAMPLE_CASE(false, false, double); break;
    case 1: LTP_SAMPLE_CASE(true, false, double); break;
    case 2: LTP_SAMPLE_CASE(false, true, double); break;
    case 3: LTP_SAMPLE_CASE(true, true, double); break;
    case 4: LTP_SAMPLE_CASE(false, false, float); break;
    case 5: LTP_SAMPLE_CASE(true, false, float); break;
    case 6: LTP_SAMPLE_CASE(false, true, float); break;
    default: LTP_SAMPLE_CASE(true, true, float); break;
    }
#undef LTP_SAMPLE_CASE
}

void launch_envelope(hipStream_t s, long long first, long long count, long long base_first, int dof, double t_sample, Limits lim, Queries in,
                     Records rec, int window, int n_windows, double* env, unsigned long long* next_item, int resident_blocks,
                     unsigned long long* probe, const unsigned long long* tables)
{
    if (count <= 0 || n_windows <= 0) return;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    long long blocks = resident_blocks > 0 ? resident_blocks : 1536;
    if (blocks > count * ngroups) blocks = count * ngroups;
    // lanes per (joint, window) task: the largest power of two <= 64 that still gives every lane of a block a task
    const long long tasks = (long long)(dof < kSampleJointGroup ? dof : kSampleJointGroup) * n_windows;
    int lg = 0;
    while (lg < 6 && (tasks << (lg + 1)) <= kSampleThreads && (2 << lg) <= window) ++lg;
    if (probe)
        hipLaunchKernelGGL((k_envelope<true, false>), dim3((unsigned)blocks), dim3(kSampleThreads), 0, s, first, count, base_first, dof, t_sample, lim, in,
                           rec, window, n_windows, lg, env, next_item, probe, tables);
    else if (tables)
        hipLaunchKernelGGL((k_envelope<false, true>), dim3((unsigned)blocks), dim3(kSampleThreads), 0, s, first, count, base_first, dof, t_sample, lim, in,
                           rec, window, n_windows, lg, env, next_item, probe, tables);
    else
        hipLaunchKernelGGL((k_envelope<false, false>), dim3((unsigned)blocks), dim3(kSampleThreads), 0, s, first, count, base_first, dof, t_sample, lim, in,
                           rec, window, n_windows, lg, env, next_item, probe, tables);
}

void launch_replan_states(hipStream_t s, long long first, long long count, int dof, RowSpec rows, Queries in, Records rec,
                          const unsigned long long* offsets, const void* tile, bool f32, unsigned long long capacity,
                          const int* sample_index, int uniform_index,
                          double* q_0, double* v_0, double* a_0, long long sq, long long sj)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (f32)
        hipLaunchKernelGGL(k_replan_states<float>, grid, block, 0, s, first, count, dof, rows, in, rec, offsets,
                           (const float*)tile, capacity, sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
    else
        hipLaunchKernelGGL(k_replan_states<double>, grid, block, 0, s, first, count, dof, rows, in, rec, offsets,
                           (const double*)tile, capacity, sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
}

void launch_end_limit(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    hipLaunchKernelGGL(k_end_limit, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, first, count, dof, t_sample, lim, in, rec);
}

void launch_state_at(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in,
                     Records rec, const int* sample_index, int uniform_index, double* q_0, double* v_0, double* a_0,
                     long long sq, long long sj, int semantics)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    if (semantics == kSemMatlab)
        hipLaunchKernelGGL(k_state_at<kSemMatlab>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, first, count, dof, t_sample, lim, in, rec,
                           sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
    else
        hipLaunchKernelGGL(k_state_at<kSemCpp>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, first, count, dof, t_sample, lim, in, rec,
                           sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
}

}  // namespace ltp
