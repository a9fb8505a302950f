// ltp_plan_small.hip — the whole of planTrajectory for a handful of queries in ONE launch (BASELINE.json configs[0]), gfx950.
#include "ltp_sampler_lds.hpp"

namespace ltp {

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
    // A call of at most kSmallInline pairs carries its queries IN the kernel arguments: in[] points to host memory, so reading it is
    // a PCIe round trip (~2 us) that can only start once the pointer itself has arrived with the kernel arguments — a second one.
    int n_inline;                // pairs held in inl (0: read in[])
    double inl[4][16];           // q_goal, q_0, v_0, a_0 of pair i at inl[k][i]
};
constexpr int kSmallInline = 16;

struct SmallShared {             // LDS of one small-batch block
    SegTable tab;
    double t_opt[kSmallPairs][7], t_scaled[kSmallPairs][7], dir[kSmallPairs], vd[kSmallPairs];
    double treq[kSmallPairs];
    signed char mod[kSmallPairs];
    int flags[kSmallPairs], slowest[kSmallPairs], len[kSmallPairs], status[kSmallPairs];
    unsigned long long off[kSmallPairs + 1];
    int fit;
    // candidate 2 of timeScaling as the second wave evaluated it beside the first wave's candidate 1 (at most 64 pairs)
    double c2_ts[64][7], c2_vd[64];
    int c2_mod[64], c2_acc[64];
    double in0[3][kSmallPairs];     // q_0, v_0, a_0 of every pair, for the table build (not read from host memory a second time)
    unsigned long long tick[8];     // diagnostic: wall clock of thread 0 at the phase boundaries
};

// GIVEN: the switching-time records are inputs (LongTermPlanner::getTrajectory, cc:706-841: t_scaled, dir, mod, v_drive
// from io.rec, start states from io.in[1..3]); stages 1-3 are skipped, lengths are those of k_finalize.
// Every thread of every block of the grid calls this (it contains block barriers); on return the block's part is done and,
// in the last block to finish, *io.done has been set.
// SEM: kSemCpp, or kSemCpp | kPowLibm (the pow rule; the kernel exists for the C++ semantics only).
template <bool GIVEN, int SEM>
LTP_DEV void plan_small_body(int n, int dof, double t_sample, int goal_check, RowSpec rows, const Limits& lim, const SmallHost& io,
                             SmallShared& sh)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();        // the block's LDS copy of glibc's pow tables (ltp_libm_pow.hpp)
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
    // With at most 64 pairs (the single call: 7) wave 1 shadows wave 0's lanes: it loads the same inputs and, while wave 0 evaluates
    // timeScaling's first candidate, evaluates the second (cc:408-446) for the same (query, joint) — the two are independent
    // computations, the reference merely tries them in order, and three of ten joints need the second (one wave would run both, one
    // after the other: 5.6 us of a 27 us call). "First accepted in order" is then a lookup, as in k_scaling_slow.
    const bool dual = !GIVEN && pairs <= 64;
    const bool shadow = dual && t >= 64 && t < 64 + pairs;
    const int pid = shadow ? t - 64 : t;
    const bool pair = t < pairs;
    const int q = (pair || shadow) ? pid / dof : 0, j = (pair || shadow) ? pid - q * dof : 0;
    JointLimits L = {0.0, 0.0, 0.0, 0.0, 0.0, {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
    double qg = 0.0, q0 = 0.0, v0 = 0.0, a0 = 0.0;
    if constexpr (GIVEN) {
        if (t < n) { s_status[t] = 0; s_len[t] = 0; s_treq[t] = 0.0; s_slowest[t] = -1; }
        __syncthreads();
        if (pair) {
            L = load_limits(lim, j);
            q0 = io.in[1][pid]; v0 = io.in[2][pid]; a0 = io.in[3][pid];
            sh.in0[0][pid] = q0; sh.in0[1][pid] = v0; sh.in0[2][pid] = a0;
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
    if (pair || shadow) {
        L = load_limits(lim, j);
        if (io.n_inline >= pairs) { const int e = pid & (kSmallInline - 1); qg = io.inl[0][e]; q0 = io.inl[1][e]; v0 = io.inl[2][e]; a0 = io.inl[3][e]; }
        else { qg = io.in[0][pid]; q0 = io.in[1][pid]; v0 = io.in[2][pid]; a0 = io.in[3][pid]; }
    }
    if (pair) {
        sh.in0[0][pid] = q0; sh.in0[1][pid] = v0; sh.in0[2][pid] = a0;
        int flags = check_inputs_joint<SEM>(L, q0, v0, a0) ? 0 : kStatusInvalidInput;
        if (goal_check && !(qg >= L.q_min && qg <= L.q_max)) flags |= kStatusGoalOutside;
        double tt[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double dir = 0.0;
        int mod = 0;
        MatlabCtx mc;
        if (opt_switch_times<true, SEM, true>(L.a_max, L.j_max, L.v_max, L.pw, t_sample, qg, q0, v0, a0, L.v_max, tt, dir, mod, mc) == kOptFalse) flags |= kStatusOptFailed;
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
    if (shadow) {
        // wave 1: candidate 2 for the lanes wave 0 evaluates candidate 1 for
        int acc = 0, mod = 0;
        double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double vd = 0.0;
        if (s_status[q] == 0 && j != s_slowest[q]) {
            const double dir = s_dir[pid];
            double v0m = v0, a0m = a0;
            if (dir < 0.0) { v0m = -v0m; a0m = -a0m; }   // cc:372-375
            MatlabCtx mc;
            acc = scaling_case<2, SEM>(L, t_sample, qg, q0, v0m, a0m, dir, s_treq[q], vd, ts, mod, mc) ? 1 : 0;
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) sh.c2_ts[pid][k] = ts[k];
        sh.c2_vd[pid] = vd;
        sh.c2_mod[pid] = mod;
        sh.c2_acc[pid] = acc;
    }
    bool c1_rejected = false;
    double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    double vd = L.v_max;
    int mod = 0;                                       // failed query: zero record, never sampled
    if (pair && dual && s_status[q] == 0 && j != s_slowest[q]) {
        const double dir = s_dir[pid];
        double v0m = v0, a0m = a0;
        if (dir < 0.0) { v0m = -v0m; a0m = -a0m; }
        MatlabCtx mc;
        c1_rejected = !scaling_case<1, SEM>(L, t_sample, qg, q0, v0m, a0m, dir, s_treq[q], vd, ts, mod, mc);
    }
    if (dual) __syncthreads();
    if (pair) {
        const int flags = s_status[q];
        if (flags == 0) {
            if (!dual) mod = s_mod[pid];
            else if (j == s_slowest[q]) mod = s_mod[pid];
            if (j != s_slowest[q]) {
                int which = 0;
                MatlabCtx mc;
                if (!dual) {
                    time_scaling_full<SEM>(L, t_sample, qg, q0, v0, a0, s_dir[pid], s_treq[q], vd, ts, mod, which, mc);
                } else if (c1_rejected) {
                    if (sh.c2_acc[pid]) {
#pragma unroll
                        for (int k = 0; k < 7; ++k) ts[k] = sh.c2_ts[pid][k];
                        vd = sh.c2_vd[pid];
                        mod = sh.c2_mod[pid];
                    } else {
                        const double dir = s_dir[pid];
                        double v0m = v0, a0m = a0;
                        if (dir < 0.0) { v0m = -v0m; a0m = -a0m; }
                        time_scaling_tail<SEM>(L, t_sample, qg, q0, v0m, a0m, dir, s_treq[q], vd, ts, mod, which, mc);
                    }
                }
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
                    else if (k == 9) pa = sh.in0[0][rj];
                    else if (k == 10) pa = sh.in0[1][rj];
                    else if (k == 11) pa = sh.in0[2][rj];
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
        for_each_run(lim, lrec, pid, j, s_len[q], t_sample, qq, vv, aa, [](int, int, const RunCoef& rc) { return (rc.mode & kModeTail) != 0; });   // q rests in the tail
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

template <bool GIVEN, int SEM>
__global__ void __launch_bounds__(kSampleThreads)
k_plan_small(int n, int dof, double t_sample, int goal_check, RowSpec rows, Limits lim, SmallHost io)
{
    __shared__ SmallShared sh;
    plan_small_body<GIVEN, SEM>(n, dof, t_sample, goal_check, rows, lim, io, sh);
}

int small_batch_pairs() { return kSmallPairs; }
int small_batch_blocks(int dof, bool with_rows) { return !with_rows ? 1 : (dof < kSmallBlocks ? (dof > 0 ? dof : 1) : kSmallBlocks); }

void launch_plan_small(hipStream_t s, int n, int dof, double t_sample, int goal_check, RowSpec rows, Limits lim, const double* const in[4],
                       Records rec, unsigned long long* offsets, double* out_rows, unsigned long long capacity, int* end_flags,
                       unsigned int* arrivals, volatile int* done, bool records_given, bool libm_pow, bool host_inputs)
{
    SmallHost io;
    for (int k = 0; k < 4; ++k) io.in[k] = in[k];
    io.rec = rec; io.offsets = offsets; io.rows = out_rows; io.capacity = capacity; io.end_flags = end_flags; io.arrivals = arrivals;
    io.done = done;
    // the queries ride in the kernel arguments when they are few (`in` is then host memory the caller has just filled)
    io.n_inline = 0;
    if (!records_given && host_inputs && n * dof <= kSmallInline) {
        io.n_inline = n * dof;
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < n * dof; ++i) io.inl[k][i] = in[k][i];
    }
    const dim3 grid((unsigned)small_batch_blocks(dof, out_rows != nullptr));
    // (with the records given there are no powers left to form: one instantiation)
    if (records_given) hipLaunchKernelGGL((k_plan_small<true, kSemCpp>), grid, dim3(kSampleThreads), 0, s, n, dof, t_sample, goal_check, rows, lim, io);
    else if (libm_pow) hipLaunchKernelGGL((k_plan_small<false, kSemCpp | kPowLibm>), grid, dim3(kSampleThreads), 0, s, n, dof, t_sample, goal_check, rows, lim, io);
    else hipLaunchKernelGGL((k_plan_small<false, kSemCpp>), grid, dim3(kSampleThreads), 0, s, n, dof, t_sample, goal_check, rows, lim, io);
}

}  // namespace ltp
