// ltp_consumers.hip — everything that reads a planned batch without writing dense rows, gfx950: the table pass
// (k_build_tables: the packed run tables of include/ltp_run_tables.hpp, for the library's own short-row sampler / envelope consumer
// and, through ltp_build_tables_batch, for USER consumers), the envelope consumer (k_envelope), the receding-horizon restart states
// (k_state_at, k_replan_states) and the end-limit verdict without rows (k_end_limit).
#include "ltp_sampler_lds.hpp"

namespace ltp {

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
// ANALYTIC (ltp_set_envelope_mode(p, LTP_ENVELOPE_ANALYTIC), round 5): inside a run q(m) is ONE cubic in the run-local index m, so its
// extreme samples over a stretch of the run are the stretch's two end samples or the samples either side of a real root of
// q'(m) = c1 + 2 c2 m + 3 c3 m^2. The task evaluates those <= 6 candidates per (run, window) stretch — with run_eval_q, i.e. they ARE
// samples of the row, bit for bit — instead of every sample. What it can miss is a sample that undercuts its neighbour by rounding
// alone: the result is within a few ulps of q (~1e-15) of the exhaustive form's, not bit-identical, which is why the exhaustive form
// stays the default (tests: 1e-12 against the exhaustive form, 1e-9 against the CPU checker's reduced rows).
template <bool PROBE, bool TABLES, bool ANALYTIC = false>
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
        if constexpr (TABLES) install_run_tables(tab.jt, nj, regs.w, t_sample);
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
                // (through the public consumer interface, include/ltp_run_tables.hpp: JointTable in LDS, RunCursor, run_eval_q)
                const int jl = task / n_windows, w = task - jl * n_windows;
                const JointTable& jt = tab.jt[jl];
                const long long b = (long long)w * window;
                const bool past = b >= (long long)len;                            // past the end: the last sample only
                int i = past ? len - 1 + r : (int)b + r;
                const int e = (b + window < (long long)len) ? (int)(b + window) : len;
                RunCursor cu(jt);
                // the four q coefficients of the current run stay in registers; they are re-read at a run boundary only
                double c4[4] = {jt.c[0][0], jt.c[0][1], jt.c[0][2], jt.c[0][3]};
                if constexpr (ANALYTIC) {
                    // (one lane per task: the host launches this form with lg = 0)
                    const int e_task = past ? len : e;
                    auto fold = [&](int m) {
                        const double q = run_eval_q(c4, m);
                        lo = __builtin_fmin(lo, q);
                        hi = __builtin_fmax(hi, q);
                    };
                    while (i < e_task) {
                        if (cu.advance(jt, i)) {
#pragma unroll
                            for (int x = 0; x < 4; ++x) c4[x] = jt.c[cu.run][x];
                        }
                        const int stretch_end = cu.nxt < e_task ? cu.nxt : e_task;     // samples [i, stretch_end) lie in this run
                        const int m0 = i - cu.cur + 1, m1 = stretch_end - cu.cur;      // their run-local positions m0 .. m1
                        fold(m0);
                        if (m1 > m0) fold(m1);
                        if (m1 - m0 > 1) {
                            const double A = 3.0 * c4[3], B = 2.0 * c4[2], C = c4[1];
                            double r1 = __builtin_nan(""), r2 = r1;
                            if (A == 0.0) {
                                if (B != 0.0) r1 = -C / B;
                            } else {
                                const double disc = B * B - 4.0 * A * C;
                                if (disc >= 0.0) {
                                    const double sq = __builtin_sqrt(disc);
                                    const double qq = -0.5 * (B + (B < 0.0 ? -sq : sq));   // the cancellation-free root first
                                    r1 = qq / A;
                                    r2 = qq != 0.0 ? C / qq : r1;
                                }
                            }
#pragma unroll
                            for (int which = 0; which < 2; ++which) {
                                const double rho = which ? r2 : r1;
                                if (rho > (double)m0 - 1.0 && rho < (double)m1 + 1.0) {   // false for NaN
                                    const int k = (int)__builtin_floor(rho);
                                    if (k > m0 && k < m1) fold(k);
                                    if (k + 1 > m0 && k + 1 < m1) fold(k + 1);
                                }
                            }
                        }
                        i = stretch_end;
                    }
                } else
                for (; i < e; i += g) {
                    if (cu.advance(jt, i)) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) c4[x] = jt.c[cu.run][x];
                    }
                    const double q = run_eval_q(c4, i - cu.cur + 1);
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

// The analytic envelopes WITHOUT run tables (round 5): lane = (plan, joint) walks its runs in registers — the walk of k_state_at /
// k_end_limit — and while a run lasts folds the candidates of each window it crosses: the two end samples of the stretch and the
// samples either side of the real roots of q'(m) (computed once per run: they do not depend on the window), evaluated with
// run_eval_q on the coefficients for_each_run hands over, i.e. the values k_envelope's analytic form folds, bit for bit. A window
// is stored (16 bytes) when the walk leaves it; the walk goes on to the last sample, which also gives planTrajectory's end-limit
// verdict (cc:59-61) and the content of the windows past the end. No table pass, no workspace, no LDS, no per-item latency: what
// the block-cooperative kernel spends on fetching and installing seven joints' tables per plan (E7.4) is not there.
template <int SEM>
__global__ void __launch_bounds__(256)
k_envelope_walk(long long first, long long count, long long base_first, int dof, double t_sample, Limits lim, Queries in, Records rec,
                int window, int n_windows, double* __restrict__ env)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * dof) return;
    const long long local = idx / dof;
    const int j = (int)(idx - local * dof);
    const long long p = first + local;
    double2_t* const dst = reinterpret_cast<double2_t*>(env) + ((unsigned long long)(p - base_first) * dof + j) * n_windows;
    const int len = rec.traj_len[p];
    if (len <= 0) {
        const double nan = __builtin_nan("");
        for (int w = 0; w < n_windows; ++w) dst[w] = double2_t{nan, nan};
        return;
    }
    const long long ix = p * in.sq + (long long)j * in.sj;
    double q = in.q_0[ix], v = in.v_0[ix], a = in.a_0[ix];
    int w = 0;                                                  // the window under construction: samples [w_end - window, w_end)
    long long w_end = window;
    double lo = __builtin_huge_val(), hi = -__builtin_huge_val();
    for_each_run<SEM>(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [&](int b, int e, const RunCoef& rc) {
        if (w >= n_windows) return false;                       // every window written: the walk only continues for the end-limit verdict
        const double* c4 = rc.c;                                // q(m) = c0 + c1 m + c2 m^2 + c3 m^3, m = sample - b + 1 (run_eval_q)
        auto fold = [&](int m) {
            const double x = run_eval_q(c4, m);
            lo = __builtin_fmin(lo, x);
            hi = __builtin_fmax(hi, x);
        };
        // the real roots of q'(m) = c1 + 2 c2 m + 3 c3 m^2 (the same expressions as k_envelope's analytic form)
        const double A = 3.0 * c4[3], B = 2.0 * c4[2], C = c4[1];
        double r1 = __builtin_nan(""), r2 = r1;
        if (A == 0.0) {
            if (B != 0.0) r1 = -C / B;
        } else {
            const double disc = B * B - 4.0 * A * C;
            if (disc >= 0.0) {
                const double sq = __builtin_sqrt(disc);
                const double qq = -0.5 * (B + (B < 0.0 ? -sq : sq));
                r1 = qq / A;
                r2 = qq != 0.0 ? C / qq : r1;
            }
        }
        int i = b;
        while (i < e && w < n_windows) {
            const int stretch_end = (long long)e < w_end ? e : (int)w_end;      // samples [i, stretch_end) of this run lie in window w
            const int m0 = i - b + 1, m1 = stretch_end - b;
            fold(m0);
            if (m1 > m0) fold(m1);
            if (m1 - m0 > 1) {
#pragma unroll
                for (int which = 0; which < 2; ++which) {
                    const double rho = which ? r2 : r1;
                    if (rho > (double)m0 - 1.0 && rho < (double)m1 + 1.0) {     // false for NaN
                        const int k = (int)__builtin_floor(rho);
                        if (k > m0 && k < m1) fold(k);
                        if (k + 1 > m0 && k + 1 < m1) fold(k + 1);
                    }
                }
            }
            i = stretch_end;
            if ((long long)i == w_end) {                                        // the window is complete
                dst[w] = double2_t{lo, hi};
                ++w;
                w_end += window;
                lo = __builtin_huge_val();
                hi = -__builtin_huge_val();
            }
        }
        return false;
    }, j == dof - 1);
    // q now holds the last trajectory sample. A window the trajectory ended in holds the samples that exist; windows that start at or
    // after the end hold the last position twice (as k_envelope)
    if (w < n_windows && lo <= hi) { dst[w] = double2_t{lo, hi}; ++w; }
    for (; w < n_windows; ++w) dst[w] = double2_t{q, q};
    if constexpr (SEM == kSemCpp) {
        if (q < lim.q_min[j] || q > lim.q_max[j]) atomicOr(&rec.status[p], kStatusEndLimit);   // cc:59-61
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

// ltp_replan_states_batch for float64 tiles: the restart state = STORED sample k of the rows ltp_sample_batch wrote. Those rows hold
// run_eval(run_coef(..)) of the run walk below, bit for bit, so the state is recomputed from the records (k_state_at's walk, ~0.1 ms
// per 1 M plans) instead of being gathered from the tile by 21 M scattered 8-byte reads (0.385 ms, at the rate of DRAM sectors):
// the tile is not read at all. Same rules as k_replan_states for plans the sampler skipped (no trajectory, LTP_STATUS_OVERFLOW,
// rows that would end beyond the tile): they keep their start state.
template <int SEM>
__global__ void __launch_bounds__(256)
k_replan_walk(long long first, long long count, int dof, double t_sample, RowSpec rows, Limits lim, Queries in, Records rec,
              const unsigned long long* __restrict__ offsets, unsigned long long capacity, const int* __restrict__ sample_index, int uniform_index,
              double* __restrict__ q_0, double* __restrict__ v_0, double* __restrict__ a_0, long long sq, long long sj)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * dof) return;
    const long long local = idx / dof;
    const int j = (int)(idx - local * dof);
    const long long p = first + local;
    const long long dst = local * sq + (long long)j * sj;
    const long long ix = p * in.sq + (long long)j * in.sj;
    double q = in.q_0[ix], v = in.v_0[ix], a = in.a_0[ix];
    const int len = rec.traj_len[p];
    const int slen = stored_len(len, rows);
    const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    const unsigned long long rel = offsets[p] - offsets[first];
    if (slen > 0 && !(rec.status[p] & kStatusOverflow) && rel + 4ull * dof * stride <= capacity) {
        int k = sample_index ? sample_index[local] : uniform_index;
        k = k < 0 ? 0 : (k >= slen ? slen - 1 : k);               // beyond the stored samples: the last stored state
        const int kt = k * (rows.stride > 1 ? rows.stride : 1);   // stored sample k is trajectory sample k * stride (< len)
        for_each_run<SEM>(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [&](int b, int e, const RunCoef& rc) {
            if (kt >= e) return false;
            double jj;
            run_eval(rc.c, kt + 1 - b, q, v, a, jj);
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
    // (a tail run, i > s6, has a = v = 0: q no longer moves, so the state before the first of them IS the last sample's position — the
    // two or three one-sample runs behind s6 need not be walked)
    for_each_run(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [](int, int, const RunCoef& rc) { return (rc.mode & kModeTail) != 0; });
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

int envelope_resident_blocks(int device)
{
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_envelope<false, false>, kSampleThreads, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 4;
    return cus * per_cu;
}

unsigned long long table_bytes(long long lanes) { return run_table_bytes(lanes); }

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

void launch_envelope(hipStream_t s, long long first, long long count, long long base_first, int dof, double t_sample, Limits lim, Queries in,
                     Records rec, int window, int n_windows, double* env, unsigned long long* next_item, int resident_blocks,
                     unsigned long long* probe, const unsigned long long* tables, bool analytic)
{
    if (count <= 0 || n_windows <= 0) return;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    long long blocks = resident_blocks > 0 ? resident_blocks : 1536;
    if (blocks > count * ngroups) blocks = count * ngroups;
    // lanes per (joint, window) task: the largest power of two <= 64 that still gives every lane of a block a task
    const long long tasks = (long long)(dof < kSampleJointGroup ? dof : kSampleJointGroup) * n_windows;
    int lg = 0;
    while (lg < 6 && (tasks << (lg + 1)) <= kSampleThreads && (2 << lg) <= window) ++lg;
    if (analytic && !probe) {
        // the analytic form looks at a handful of samples per run and window: one lane per (joint, window) task
        if (tables)
            hipLaunchKernelGGL((k_envelope<false, true, true>), dim3((unsigned)blocks), dim3(kSampleThreads), 0, s, first, count, base_first, dof, t_sample, lim, in,
                               rec, window, n_windows, 0, env, next_item, probe, tables);
        else
            hipLaunchKernelGGL((k_envelope<false, false, true>), dim3((unsigned)blocks), dim3(kSampleThreads), 0, s, first, count, base_first, dof, t_sample, lim, in,
                               rec, window, n_windows, 0, env, next_item, probe, tables);
        return;
    }
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

void launch_envelope_walk(hipStream_t s, long long first, long long count, long long base_first, int dof, double t_sample, Limits lim, Queries in,
                          Records rec, int window, int n_windows, double* env, int semantics)
{
    if (count <= 0 || n_windows <= 0 || dof <= 0) return;
    const long long total = count * dof;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (semantics == kSemMatlab)
        hipLaunchKernelGGL(k_envelope_walk<kSemMatlab>, grid, block, 0, s, first, count, base_first, dof, t_sample, lim, in, rec, window, n_windows, env);
    else
        hipLaunchKernelGGL(k_envelope_walk<kSemCpp>, grid, block, 0, s, first, count, base_first, dof, t_sample, lim, in, rec, window, n_windows, env);
}

void launch_replan_states(hipStream_t s, long long first, long long count, int dof, RowSpec rows, Queries in, Records rec,
                          const unsigned long long* offsets, const void* tile, bool f32, unsigned long long capacity,
                          const int* sample_index, int uniform_index,
                          double* q_0, double* v_0, double* a_0, long long sq, long long sj, double t_sample, Limits lim, int semantics)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (f32)      // float rows hold ROUNDED values: the caller gets what the tile holds
        hipLaunchKernelGGL(k_replan_states<float>, grid, block, 0, s, first, count, dof, rows, in, rec, offsets,
                           (const float*)tile, capacity, sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
    else if (semantics == kSemMatlab)
        hipLaunchKernelGGL(k_replan_walk<kSemMatlab>, grid, block, 0, s, first, count, dof, t_sample, rows, lim, in, rec, offsets, capacity, sample_index,
                           uniform_index, q_0, v_0, a_0, sq, sj);
    else
        hipLaunchKernelGGL(k_replan_walk<kSemCpp>, grid, block, 0, s, first, count, dof, t_sample, rows, lim, in, rec, offsets, capacity, sample_index,
                           uniform_index, q_0, v_0, a_0, sq, sj);
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
