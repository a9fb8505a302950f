// ltp_sampler_lds.hpp — the run tables of one (plan, joint group) item in LDS, built by the whole block (build_run_tables), and
// the row writer that streams an item's rows from them (stream_rows). Shared by k_sample (ltp_sampler.hip), k_envelope
// (ltp_consumers.hip) and k_plan_small (ltp_plan_small.hip).
#pragma once
#include "ltp_runs.hpp"

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
struct SegTable {
    JointTable jt[kSampleJointGroup];     // include/ltp_run_tables.hpp
    union {
        SegScratch w;
        // the sampler reuses the space for the finished 16-byte slots that contain run boundary k: [q, v, a, j]
        double2_t bnd[kSampleJointGroup][kMaxSegments][4];
    };
};
static_assert(kSampleJointGroup == kRunTableJoints && kSampleThreads == kRunTableThreads, "install_run_tables is written for this block shape");

// candidate cut points: slot 0 is index 0, slot c >= 1 is s[kCutBase[c]] + kCutDelta[c]; every index where the jerk
// array or a snap rule (cc:815-829) can change is among them
constexpr int kCutSlots = 20;
__device__ const signed char kCutBase[kCutSlots] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 6};
__device__ const signed char kCutDelta[kCutSlots] = {0, 0, 1, 2, 0, 1, 0, 1, 2, -1, 0, 1, 0, 1, 2, 0, 1, 0, 1, 2};

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
        const PackedTableRegs t = fetch_run_tables(tables, (unsigned long long)(p - tab_first) * dof + j0, nj);
#pragma unroll
        for (int x = 0; x < kTableLoads; ++x) r.w[x] = t.w[x];
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

}  // namespace ltp
