// ltp_stage_kernels.hip — stages 1-3 of planTrajectory (switching times) and the packed-offset scan, gfx950.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile). No fast-math:
// the inf/NaN flow of the reference is part of the contract.
#include "ltp_device.hpp"

namespace ltp {

// ---------------------------------------------------------------------------------------
// Stages 1-3 of planTrajectory (cc:14-55) as four kernels:
//   k_opt_fast      every (query, joint) lane: checkInputs + optSwitchTimes(v_max) WITHOUT the quartic sites;
//                   lanes that reach them are compacted into queue A
//   k_opt_slow      queue A, densely: optSwitchTimes with the root finder
//   k_reduce_scale  per query: slowest-joint reduction through LDS (cc:31-39), then timeScaling cases c1/c2
//                   (closed form) per lane; lanes that need c3..c8 or hit a quartic site go to queue B
//   k_scaling_slow  queue B, densely: all eight cases in order + reset + fallback
// The two "fast" kernels carry no polynomial solver (fewer registers, small code); the rare, expensive and
// divergent paths run with full waves instead of dragging 64-lane waves of the main kernels through them.
//
// Block = 64 queries x JB joint slots; wave y handles joints y, y+JB, ... of 64 consecutive queries, so the
// joint limits are wave-uniform (SGPRs) and both input layouts are read with one stride per lane.
// ---------------------------------------------------------------------------------------
constexpr int kLaneGoalOutside = 128; // lane_flags bit: q_goal outside [q_min, q_max] (only with the opt-in goal check)
constexpr int kLaneDeferred = 64;   // lane_flags bit: optSwitchTimes of this lane is still pending in queue A
// MATLAB semantics only: the lane's MatlabCtx flags (complex intermediate / LTPlanner.m would have raised an error), mapped to
// kStatusMatlabComplex / kStatusMatlabError when the per-query status is formed
constexpr int kLaneMatlabComplex = 16, kLaneMatlabError = 32;
LTP_DEV int matlab_lane_bits(const MatlabCtx& mc)
{
    return ((mc.flags & kMatlabComplex) ? kLaneMatlabComplex : 0) | ((mc.flags & kMatlabError) ? kLaneMatlabError : 0);
}
LTP_DEV int matlab_status_bits(const MatlabCtx& mc)
{
    return ((mc.flags & kMatlabComplex) ? kStatusMatlabComplex : 0) | ((mc.flags & kMatlabError) ? kStatusMatlabError : 0);
}

// Compaction queues. A single device-scope counter saturates near 90 atomics/us on MI355X, which a kernel that
// pushes from ~10^5 waves would run into; so a queue has kQueueShards segments with one counter each (shard =
// blockIdx & 7, i.e. the blocks that share an XCD under round-robin dispatch), and a block aggregates its waves'
// ballots in LDS and issues ONE atomicAdd per push round.
constexpr int kQueueShards = 8;

struct Queue {
    unsigned long long* items;    // kQueueShards segments of `segment` entries
    unsigned long long* counts;   // [kQueueShards]
    unsigned long long segment;
};

// Must be called by every thread of a (64, JB) block (contains barriers). s_cnt: >= kMaxJointSlots + 1 words of LDS.
LTP_DEV void block_push(bool want, unsigned long long item, const Queue& Q, unsigned long long* s_cnt)
{
    const int lane = threadIdx.x, wave = threadIdx.y, nw = blockDim.y;
    const int shard = blockIdx.x & (kQueueShards - 1);
    const unsigned long long mask = __ballot(want);
    if (lane == 0) s_cnt[wave] = (unsigned long long)__popcll(mask);
    __syncthreads();
    if (wave == 0 && lane == 0) {
        unsigned long long total = 0ull;
        for (int w = 0; w < nw; ++w) total += s_cnt[w];
        s_cnt[kMaxJointSlots] = total ? atomicAdd(&Q.counts[shard], total) : 0ull;
    }
    __syncthreads();
    if (want) {
        unsigned long long off = s_cnt[kMaxJointSlots];
        for (int w = 0; w < wave; ++w) off += s_cnt[w];
        off += (unsigned long long)__popcll(mask & ((1ull << lane) - 1ull));
        Q.items[(unsigned long long)shard * Q.segment + off] = item;
    }
    __syncthreads();
}

// item `it` of the concatenated shards (it < queue_total)
LTP_DEV unsigned long long queue_item(const Queue& Q, const unsigned long long (&cnt)[kQueueShards], unsigned long long it)
{
    int sh = 0;
#pragma unroll
    for (int k = 0; k < kQueueShards - 1; ++k) {
        if (sh == k && it >= cnt[k]) { it -= cnt[k]; sh = k + 1; }
    }
    return Q.items[(unsigned long long)sh * Q.segment + it];
}

LTP_DEV unsigned long long queue_total(const Queue& Q, unsigned long long (&cnt)[kQueueShards])
{
    unsigned long long total = 0ull;
#pragma unroll
    for (int k = 0; k < kQueueShards; ++k) { cnt[k] = Q.counts[k]; total += cnt[k]; }
    return total;
}

// cc:50-55: max_element(t_scaled) <= 0 (a NaN in the first place wins every comparison of the scan, as in std::max_element);
// LTPlanner.m:82: ~any(t_scaled), i.e. every entry exactly zero (NaN counts as non-zero)
template <int SEM>
LTP_DEV bool needs_fallback(const double (&ts)[7])
{
    if constexpr (sem_matlab(SEM)) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < 7; ++k) any = any || ts[k] != 0.0;
        return !any;
    } else {
        double mx = ts[0];
#pragma unroll
        for (int k = 1; k < 7; ++k) if (mx < ts[k]) mx = ts[k];
        return mx <= 0.0;
    }
}

LTP_DEV void store_opt_record(const Records& out, long long rj, const double (&t)[7], double dir, int mod)
{
#pragma unroll
    for (int k = 0; k < 7; ++k) out.t_opt[rj * 7 + k] = t[k];
    out.dir[rj] = dir;
    out.mod[rj] = (signed char)mod;
}

template <int SEM>
__global__ void __launch_bounds__(kQueriesPerBlock* kMaxJointSlots)
k_opt_fast(long long n, int dof, double t_sample, int goal_check, Limits lim, Queries in, Records out,
           signed char* __restrict__ lane_flags, Queue queue)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();        // the block's LDS copy of glibc's pow tables (ltp_libm_pow.hpp)
    __shared__ unsigned long long s_cnt[kMaxJointSlots + 1];
    const int x = threadIdx.x, y = threadIdx.y, JB = blockDim.y;
    const long long q = (long long)blockIdx.x * kQueriesPerBlock + x;
    const bool live = q < n;
    // every wave runs the same number of rounds: block_push() contains barriers
    for (int jb = 0; jb < dof; jb += JB) {
        const int j = jb + y;
        const bool active = live && j < dof;
        const JointLimits L = load_limits(lim, j < dof ? j : dof - 1);
        const long long rj = q * dof + j;
        bool defer = false;
        if (active) {
            const long long ix = q * in.sq + (long long)j * in.sj;
            const double qg = in.q_goal[ix], q0 = in.q_0[ix], v0 = in.v_0[ix], a0 = in.a_0[ix];
            int flags = check_inputs_joint<SEM>(L, q0, v0, a0) ? 0 : kStatusInvalidInput;
            // NEW, opt-in: the reference never checks q_goal (cc:68-77), only the last sample (cc:59-61)
            if (goal_check && !(qg >= L.q_min && qg <= L.q_max)) flags |= kLaneGoalOutside;
            double t[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            double dir = 0.0;
            int mod = 0;
            MatlabCtx mc;
            const int rc = opt_switch_times<false, SEM>(L.a_max, L.j_max, L.v_max, L.pw, t_sample, qg, q0, v0, a0, L.v_max, t, dir, mod, mc);
            if (rc == kOptDefer) {
                defer = true;
                flags |= kLaneDeferred;
            } else {
                if (rc == kOptFalse) flags |= kStatusOptFailed;
                if constexpr (sem_matlab(SEM)) {
                    flags |= matlab_lane_bits(mc);
                    mod = 0;   // LTPlanner.m:64 discards optSwitchTimes' third output: mod_jerk_profile stays false after stage 1
                }
                store_opt_record(out, rj, t, dir, mod);
            }
            lane_flags[rj] = (signed char)flags;
        }
        block_push(defer, (unsigned long long)rj, queue, s_cnt);
    }
}

// k_scaling_slow in MATLAB semantics: how many consecutive queue items a block takes per pass. LAPACK's iteration runs 16-30 sweeps on
// nearly every degree-6 polynomial with different windows per lane; at 2 waves per SIMD (the kernel's register budget) the machine holds
// kSlowWavesAtOnce waves, and the queue of a 100 k batch has ~1 300 items: 64 to a wave leaves 98 % of the SIMDs idle and makes every
// wave run the union of 64 lanes' branches.
constexpr unsigned kSlowWavesAtOnce = 256 * 4 * 2;
constexpr unsigned long long kSlowLanesMin = 1;   // (a floor of 8 lanes per block was measured: slower in every line, profiles/r05_matlab_roots_registers_ab.txt)
__device__ __forceinline__ unsigned long long slow_lanes_per_block(unsigned long long count, unsigned grid, unsigned blocks_at_once)
{
    const unsigned long long blocks = grid < blocks_at_once ? grid : blocks_at_once;
    const unsigned long long per = (count + blocks - 1) / blocks;
    return per < kSlowLanesMin ? kSlowLanesMin : per > kQueriesPerBlock ? kQueriesPerBlock : per;
}

template <int SEM>
__global__ void __launch_bounds__(64)
k_opt_slow(int dof, double t_sample, Limits lim, Queries in, Records out, signed char* __restrict__ lane_flags, Queue queue)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();        // the block's LDS copy of glibc's pow tables (ltp_libm_pow.hpp)
    unsigned long long cnt[kQueueShards];
    const unsigned long long count = queue_total(queue, cnt);
    // (64 queued lanes per block in both semantics. Dealt thinly to all the blocks that run at once, as k_scaling_slow does in MATLAB
    // semantics, this kernel was SLOWER — 88 -> 107 us per 100 k queries of the reference's limits in C++ semantics, 0.560 -> 0.588 ms for
    // the whole MATLAB-semantics line: its queue is long where it matters, one wave's binary64 stream fills its SIMD, and two thin waves
    // per SIMD issue twice the instructions for the same lanes.)
    for (unsigned long long it = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; it < count;
         it += (unsigned long long)gridDim.x * blockDim.x) {
        const long long rj = (long long)queue_item(queue, cnt, it);
        const long long q = rj / dof;
        const int j = (int)(rj - q * dof);
        const JointLimits L = load_limits(lim, j);
        const long long ix = q * in.sq + (long long)j * in.sj;
        double t[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double dir = 0.0;
        int mod = 0;
        MatlabCtx mc;
        const int rc = opt_switch_times<true, SEM>(L.a_max, L.j_max, L.v_max, L.pw, t_sample, in.q_goal[ix], in.q_0[ix], in.v_0[ix], in.a_0[ix],
                                                   L.v_max, t, dir, mod, mc);
        if constexpr (sem_matlab(SEM)) mod = 0;   // LTPlanner.m:64
        store_opt_record(out, rj, t, dir, mod);
        int flags = lane_flags[rj] & ~kLaneDeferred;
        if (rc == kOptFalse) flags |= kStatusOptFailed;
        if constexpr (sem_matlab(SEM)) flags |= matlab_lane_bits(mc);
        lane_flags[rj] = (signed char)flags;
    }
}

// (with glibc's pow the kernel has 155 VGPRs = one 7-wave block per compute unit; held to 128 for two blocks it spills 80 bytes and is
// 2 % slower on the same box: left alone. Same for k_opt_fast at 80 registers, -15 %.)
template <int SEM>
__global__ void __launch_bounds__(kQueriesPerBlock* kMaxJointSlots)
__attribute__((amdgpu_waves_per_eu(4, 4)))      // 128 registers (libm rule: 144 -> 128 with 32 bytes of scratch): four 4-wave blocks per compute unit, 568 -> 534 us per 1 M
k_reduce_scale(long long n, int dof, double t_sample, Limits lim, Queries in, Records out,
               const signed char* __restrict__ lane_flags, Queue queue)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();        // the block's LDS copy of glibc's pow tables (ltp_libm_pow.hpp)
    __shared__ double s_t[kMaxJointSlots][kQueriesPerBlock];
    __shared__ int s_j[kMaxJointSlots][kQueriesPerBlock];
    __shared__ int s_f[kMaxJointSlots][kQueriesPerBlock];
    __shared__ double s_treq[kQueriesPerBlock];
    __shared__ int s_len[kQueriesPerBlock], s_bad[kQueriesPerBlock];
    // lanes whose first candidate was rejected (local lane ids); (query, joint) records for queue B
    __shared__ unsigned short s_second[kMaxJointSlots * kQueriesPerBlock];
    __shared__ unsigned long long s_slow[kMaxJointSlots * kQueriesPerBlock];
    __shared__ int s_nsecond, s_nslow;
    __shared__ unsigned long long s_base;

    const int x = threadIdx.x, y = threadIdx.y, JB = blockDim.y;
    const int tid = y * kQueriesPerBlock + x;
    const long long q = (long long)blockIdx.x * kQueriesPerBlock + x;
    const bool live = q < n;

    // cc:31-39: strict '>', first index wins, NaN never wins, init -1
    double best_t = -1.0;
    int best_j = -1, flags = 0;
    if (live) {
        for (int j = y; j < dof; j += JB) {
            const long long rj = q * dof + j;
            const double t6 = out.t_opt[rj * 7 + 6];
            flags |= lane_flags[rj] & 0xff;
            if (t6 > best_t) { best_t = t6; best_j = j; }
        }
    }
    s_t[y][x] = best_t;
    s_j[y][x] = best_j;
    s_f[y][x] = flags;
    if (tid == 0) { s_nsecond = 0; s_nslow = 0; }
    __syncthreads();
    double t_required = -1.0;
    int slowest = -1;
    flags = 0;
    for (int yy = 0; yy < JB; ++yy) {
        const double bt = s_t[yy][x];
        const int bj = s_j[yy][x];
        flags |= s_f[yy][x];
        if (bj >= 0 && (bt > t_required || (bt == t_required && bj < slowest))) { t_required = bt; slowest = bj; }
    }
    if (slowest < 0) flags |= kStatusNoSlowest;
    if (flags & kLaneGoalOutside) flags = (flags & ~kLaneGoalOutside) | kStatusGoalOutside;
    if constexpr (sem_matlab(SEM)) {
        // lane bits -> status bits, after the goal-outside bit has moved (its lane number is kStatusMatlabError's; the two MATLAB
        // lane bits share numbers with NONFINITE / OVERFLOW, which nothing has set yet)
        const int mb = ((flags & kLaneMatlabComplex) ? kStatusMatlabComplex : 0) | ((flags & kLaneMatlabError) ? kStatusMatlabError : 0);
        flags = (flags & ~(kLaneMatlabComplex | kLaneMatlabError)) | mb;
    }
    // whether the query is planned at all: kStatusMatlabComplex is informational (the plan is delivered)
    const bool planned = (flags & ~kStatusMatlabComplex) == 0;
    if (y == 0) {
        s_treq[x] = t_required;
        s_len[x] = 0;
        s_bad[x] = 0;
        if (live) {
            out.t_required[q] = t_required;
            out.slowest[q] = slowest;
            out.status[q] = flags;
        }
    }
    __syncthreads();

    // cc:43-55 with the closed-form candidates c1, c2 (cc:378-446). A finished (query, joint) lane stores its record and
    // folds its length into s_len / s_bad (traj_len, cc:716-719; queue-B lanes add theirs with atomicMax later).
    auto finish = [&](long long rj, int col, bool valid, double (&ts)[7], double vd, int mod) {
        if (valid) {
            // cc:50-55: no scaled solution (or the slowest joint) -> optimal times
            if (needs_fallback<SEM>(ts)) {
#pragma unroll
                for (int k = 0; k < 7; ++k) ts[k] = out.t_opt[rj * 7 + k];
            }
        }
#pragma unroll
        for (int k = 0; k < 7; ++k) out.t_scaled[rj * 7 + k] = ts[k];
        out.v_drive[rj] = vd;
        out.mod[rj] = (signed char)mod;
        if (valid) {
            const int l = joint_len(ts, t_sample);
            if (l < 0) atomicOr(&s_bad[col], 1);
            else atomicMax(&s_len[col], l);
        }
    };
    for (int jb = 0; jb < dof; jb += JB) {   // same number of rounds in every wave: the loop contains barriers
        // (1) the first candidate, lane = (query, joint), limits wave-uniform. 85 % of the scaled joints end here; a lane
        //     whose c1 is rejected only leaves its id, so that the second candidate — twenty divisions and another
        //     optSwitchTimes, which every wave would otherwise execute for the sake of a few of its lanes — is evaluated
        //     by full waves in (2).
        const int j = jb + y;
        const bool active = live && j < dof;
        const JointLimits L = load_limits(lim, j < dof ? j : dof - 1);
        const long long rj = q * dof + j;
        if (active) {
            double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            double vd = L.v_max;
            int mod = 0;   // failed query: zero record, never sampled
            int acc = kOptTrue;
            MatlabCtx mc;
            if (planned) {
                mod = out.mod[rj];
                if (j != slowest) {
                    const long long ix = q * in.sq + (long long)j * in.sj;
                    const double qg = in.q_goal[ix], q0 = in.q_0[ix];
                    double v0 = in.v_0[ix], a0 = in.a_0[ix];
                    const double dir = out.dir[rj];
                    if (dir < 0.0) { v0 = -v0; a0 = -a0; }
                    vd = v_drive_candidate<1, SEM>(L.a_max, L.j_max, L.pw, qg, q0, v0, a0, dir, t_required, mc);
                    acc = try_v_drive<false, SEM>(L.a_max, L.j_max, L.v_max, L.pw, t_sample, qg, q0, v0, a0, dir, t_required, vd, ts, mod, mc);
                }
            }
            if constexpr (sem_matlab(SEM)) {
                if (mc.flags && acc != kOptDefer) atomicOr(&out.status[q], matlab_status_bits(mc));
            }
            if (acc == kOptTrue) finish(rj, x, planned, ts, vd, mod);
            else if (acc == kOptFalse) s_second[atomicAdd(&s_nsecond, 1)] = (unsigned short)tid;
            else s_slow[atomicAdd(&s_nslow, 1)] = (unsigned long long)rj;      // c1 reached a quartic site: all of it in queue B
        }
        __syncthreads();
        // (2) the second candidate for the lanes that need it, densely: thread e takes the e-th such lane (limits per lane)
        const int nsecond = s_nsecond;
        if (tid < nsecond) {
            const int who = s_second[tid];
            const int x2 = who & (kQueriesPerBlock - 1), j2 = jb + who / kQueriesPerBlock;
            const long long q2 = (long long)blockIdx.x * kQueriesPerBlock + x2;
            const long long rj2 = q2 * dof + j2;
            const JointLimits L2 = load_limits(lim, j2);
            const long long ix = q2 * in.sq + (long long)j2 * in.sj;
            const double qg = in.q_goal[ix], q0 = in.q_0[ix];
            double v0 = in.v_0[ix], a0 = in.a_0[ix];
            const double dir = out.dir[rj2], tr = s_treq[x2];
            if (dir < 0.0) { v0 = -v0; a0 = -a0; }
            double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            int mod = 0;
            MatlabCtx mc;
            const double vd = v_drive_candidate<2, SEM>(L2.a_max, L2.j_max, L2.pw, qg, q0, v0, a0, dir, tr, mc);
            const int acc = try_v_drive<false, SEM>(L2.a_max, L2.j_max, L2.v_max, L2.pw, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc);
            if constexpr (sem_matlab(SEM)) {
                if (mc.flags && acc != kOptDefer) atomicOr(&out.status[q2], matlab_status_bits(mc));
            }
            if (acc == kOptTrue) finish(rj2, x2, true, ts, vd, mod);
            else s_slow[atomicAdd(&s_nslow, 1)] = (unsigned long long)rj2;
        }
        __syncthreads();
        // (3) what neither closed form settled goes to queue B: one reservation per block and round
        const int nslow = s_nslow;
        if (nslow > 0) {
            const int shard = blockIdx.x & (kQueueShards - 1);
            if (tid == 0) s_base = atomicAdd(&queue.counts[shard], (unsigned long long)nslow);
            __syncthreads();
            if (tid < nslow) queue.items[(unsigned long long)shard * queue.segment + s_base + tid] = s_slow[tid];
        }
        __syncthreads();
        if (tid == 0) { s_nsecond = 0; s_nslow = 0; }
        __syncthreads();
    }
    if (live && y == 0) {
        out.traj_len[q] = planned ? s_len[x] : 0;
        if (s_bad[x]) atomicOr(&out.status[q], kStatusNonFinite);
    }
}

// Queue B. Block = 64 queued (query, joint) items x 8 waves; wave c evaluates candidate c+1 for all 64 items, so
// the eight candidates of cc:378-638 (independent computations) run side by side and the kernel's latency is the
// slowest candidate (the degree-6 solve) instead of their sum. The reference's "first accepted in order" is then a
// lookup over eight flags in LDS.
template <int SEM>
__global__ void __launch_bounds__(kQueriesPerBlock * 8)
k_scaling_slow(int dof, double t_sample, Limits lim, Queries in, Records out, Queue queue, int exp_per)
{
    if constexpr (sem_libm(SEM)) libm::stage_tables();        // the block's LDS copy of glibc's pow tables (ltp_libm_pow.hpp)
    __shared__ int s_acc[8][kQueriesPerBlock];
    const int x = threadIdx.x;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.y);
    // the degree-6 and degree-5 candidates are the kernel's latency: their waves go first on the SIMD they share with a quartic
    // candidate (profiles/r06_stage_small_ab.txt: 253 -> 242 us per 1 M queries, 154 -> 152 us per 100 k)
    if (c == 7 || c == 4) __builtin_amdgcn_s_setprio(3);
    unsigned long long cnt[kQueueShards];
    const unsigned long long count = queue_total(queue, cnt);
    // MATLAB semantics: the queued items spread over the blocks that run at once (one 8-wave block per compute unit),
    // 1 .. 64 per block — LAPACK's iteration takes 16-30 sweeps on nearly every degree-6 polynomial, so a wave of 64 pays the union of
    // 64 lanes' branches in every one of them (100 k queries: 0.476 -> 0.457 ms). C++ semantics keeps 64 per block: there ONE lane's 76
    // Francis steps decide the kernel whatever its company, and 256 blocks instead of 20 were 3 % slower (155.7 -> 160.6 us, one box).
    const unsigned long long per = sem_matlab(SEM) ? slow_lanes_per_block(count, gridDim.x, kSlowWavesAtOnce / 8) : (exp_per > 0 ? (unsigned long long)exp_per
                                     // a queue of up to 64 blocks' worth is dealt 32 to a block: the waves' company halves and the chip has the room
                                     // (S-ref 100 k: 323 -> 294 us; beyond that more instruction streams cost more than they free, r06_stage_small_ab.txt)
                                     : (count <= 4096ull ? 32ull : (unsigned long long)kQueriesPerBlock));
    for (unsigned long long base = (unsigned long long)blockIdx.x * per; base < count; base += (unsigned long long)gridDim.x * per) {
        const unsigned long long it = base + x;
        const bool live = (unsigned long long)x < per && it < count;
        bool acc = false;
        double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double vd = 0.0;
        int mod = 0, j = 0;
        long long rj = 0, q = 0;
        JointLimits L = {0.0, 0.0, 0.0, 0.0, 0.0, {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}};
        MatlabCtx mc;
        if (live) {
            rj = (long long)queue_item(queue, cnt, it);
            q = rj / dof;
            j = (int)(rj - q * dof);
            // (the joint's five limits here; its power table inside each candidate's case: per lane in this kernel, and held across
            // the switch it costs every candidate the registers of all seven entries — the MATLAB-semantics solver then spills)
            L.q_min = lim.q_min[j]; L.q_max = lim.q_max[j]; L.v_max = lim.v_max[j]; L.a_max = lim.a_max[j]; L.j_max = lim.j_max[j];
            const long long ix = q * in.sq + (long long)j * in.sj;
            const double qg = in.q_goal[ix], q0 = in.q_0[ix];
            double v0 = in.v_0[ix], a0 = in.a_0[ix];
            const double dir = out.dir[rj], tr = out.t_required[q];
            if (dir < 0.0) { v0 = -v0; a0 = -a0; }   // cc:372-375
            JointLimits Lc = L;
            switch (c) {
            case 0: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<1, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            case 1: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<2, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            case 2: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<3, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            case 3: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<4, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            case 4: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<5, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            case 5: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<6, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            case 6: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<7, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            default: Lc.pw = load_limit_powers(lim, j); acc = scaling_case<8, SEM>(Lc, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod, mc); break;
            }
        }
        s_acc[c][x] = acc ? 1 : 0;
        __syncthreads();
        if (live) {
            int first = -1;
#pragma unroll
            for (int cc = 7; cc >= 0; --cc) if (s_acc[cc][x]) first = cc;
            const bool winner = (first == c);
            const bool reset = (first < 0 && c == 0);   // cc:640-644
            if (reset) {
                mod = 0;
                zero7(ts);
                vd = L.v_max;
            }
            if constexpr (sem_matlab(SEM)) {
                // LTPlanner.m tries the candidates one after the other: what a candidate BEHIND the accepted one would have
                // raised or flagged never happens there
                if (mc.flags && (first < 0 || c <= first)) atomicOr(&out.status[q], matlab_status_bits(mc));
            }
            if (winner || reset) {
                if (needs_fallback<SEM>(ts)) {   // cc:50-55
#pragma unroll
                    for (int k = 0; k < 7; ++k) ts[k] = out.t_opt[rj * 7 + k];
                }
#pragma unroll
                for (int k = 0; k < 7; ++k) out.t_scaled[rj * 7 + k] = ts[k];
                out.v_drive[rj] = vd;
                out.mod[rj] = (signed char)mod;
                const int l = joint_len(ts, t_sample);
                if (l < 0) atomicOr(&out.status[q], kStatusNonFinite);
                else atomicMax(&out.traj_len[q], l);
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// traj_len (cc:716-719), per-plan packed size and the exclusive scan of sizes.
// Packed layout of plan p at out + offsets[p]: [array q,v,a,j][joint][row_stride] elements,
// row_stride = round_up(stored samples, 32) so that every row starts 256-B (f64) / 128-B (f32) aligned.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_finalize(long long n, int dof, double t_sample, RowSpec rows, Records rec, unsigned long long* __restrict__ block_sums)
{
    __shared__ unsigned long long s_part[256];
    const long long base = (long long)blockIdx.x * kScanBlock;
    unsigned long long local = 0ull;
    for (int e = 0; e < kScanBlock / 256; ++e) {
        const long long q = base + e * 256 + threadIdx.x;
        if (q < n) {
            int len = 0, st = rec.status[q];
            if ((st & ~kStatusMatlabComplex) == 0) {
                bool finite = true;
                for (int j = 0; j < dof; ++j) {
                    const double* t = rec.t_scaled + (q * dof + j) * 7;
                    const double tj[7] = {t[0], t[1], t[2], t[3], t[4], t[5], t[6]};
                    const int l = joint_len(tj, t_sample);
                    finite = finite && l >= 0;
                    len = l > len ? l : len;
                }
                if (!finite) { len = 0; st |= kStatusNonFinite; rec.status[q] = st; }
            }
            rec.traj_len[q] = len;
            local += plan_size(stored_len(len, rows), dof);
        }
    }
    s_part[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0];
}

// batched path: traj_len/status were already reduced by k_reduce_scale / k_scaling_slow
__global__ void __launch_bounds__(256)
k_finalize_lens(long long n, int dof, RowSpec rows, Records rec, unsigned long long* __restrict__ block_sums)
{
    __shared__ unsigned long long s_part[256];
    const long long base = (long long)blockIdx.x * kScanBlock;
    unsigned long long local = 0ull;
    for (int e = 0; e < kScanBlock / 256; ++e) {
        const long long q = base + e * 256 + threadIdx.x;
        if (q < n) {
            int len = rec.traj_len[q];
            if ((rec.status[q] & ~kStatusMatlabComplex) != 0) { len = 0; rec.traj_len[q] = 0; }   // failed or non-finite: nothing to sample
            local += plan_size(stored_len(len, rows), dof);
        }
    }
    s_part[threadIdx.x] = local;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) s_part[threadIdx.x] += s_part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = s_part[0];
}

// exclusive scan of block_sums in place, single block
__global__ void __launch_bounds__(1024)
k_scan_top(long long nb, unsigned long long* __restrict__ block_sums)
{
    __shared__ unsigned long long s[1024];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0ull;
    __syncthreads();
    for (long long base = 0; base < nb; base += 1024) {
        const long long i = base + threadIdx.x;
        const unsigned long long v = i < nb ? block_sums[i] : 0ull;
        s[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {
            const unsigned long long add = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0ull;
            __syncthreads();
            s[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < nb) block_sums[i] = carry + s[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += s[1023];
        __syncthreads();
    }
}

__global__ void __launch_bounds__(256)
k_scan_apply(long long n, int dof, RowSpec rows, const int* __restrict__ traj_len,
             const unsigned long long* __restrict__ block_sums, unsigned long long* __restrict__ offsets)
{
    __shared__ unsigned long long s[256];
    const long long base = (long long)blockIdx.x * kScanBlock;
    constexpr int E = kScanBlock / 256;
    // thread owns E consecutive plans
    unsigned long long sz[E], local = 0ull;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const long long q = base + (long long)threadIdx.x * E + e;
        sz[e] = q < n ? plan_size(stored_len(traj_len[q], rows), dof) : 0ull;
        local += sz[e];
    }
    s[threadIdx.x] = local;
    __syncthreads();
    for (int d = 1; d < 256; d <<= 1) {
        const unsigned long long add = (int)threadIdx.x >= d ? s[threadIdx.x - d] : 0ull;
        __syncthreads();
        s[threadIdx.x] += add;
        __syncthreads();
    }
    unsigned long long run = block_sums[blockIdx.x] + s[threadIdx.x] - local;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const long long q = base + (long long)threadIdx.x * E + e;
        if (q < n) {
            offsets[q] = run;
            run += sz[e];
            if (q == n - 1) offsets[n] = run;
        }
    }
}

// ---------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------
long long queue_segment(long long n, int dof)
{
    const long long nblocks = (n + kQueriesPerBlock - 1) / kQueriesPerBlock;
    return (nblocks + kQueueShards - 1) / kQueueShards * kQueriesPerBlock * (long long)dof;
}

void launch_switch_times(hipStream_t s, long long n, int dof, double t_sample, int goal_check, Limits lim, Queries in,
                         Records out, signed char* lane_flags, unsigned long long* queue_items /* 2 * 8 * queue_segment(n, dof) */,
                         unsigned long long* counts /* [16], zeroed by the caller on the same stream */, int variant)
{
    if (n <= 0) return;
    const int jb = dof < kMaxJointSlots ? dof : kMaxJointSlots;
    const dim3 block(kQueriesPerBlock, jb);
    const dim3 grid((unsigned)((n + kQueriesPerBlock - 1) / kQueriesPerBlock));
    const unsigned long long seg = (unsigned long long)queue_segment(n, dof);
    const Queue qa{queue_items, counts, seg};
    const Queue qb{queue_items + kQueueShards * seg, counts + kQueueShards, seg};
    // queue lengths are only known on the device: fixed grids, grid-stride over the queues
    long long a_blocks = (n * dof + 63) / 64;
    if (a_blocks > 4096) a_blocks = 4096;
    long long b_blocks = (n * dof + kQueriesPerBlock - 1) / kQueriesPerBlock;
    if (b_blocks > 1024) b_blocks = 1024;
    // Joint slots per block. Under the libm pow rule k_reduce_scale holds 144 and k_opt_fast 108 registers per lane (3 and 4 waves per
    // SIMD): 64 x 4 blocks fill those slots (three / four blocks per compute unit), 64 x 7 blocks leave 5 of 12 and 2 of 16 empty
    // (profiles/r06_stage_block_shape_ab.txt: 971 -> 674 us and 375 -> 341 us per 1 M 7-DoF plans; with the exact rule's
    // smaller kernels the one-round 64 x dof block stays ahead, 448 vs 461 us).
    // (the A/B runs of profiles/r06_stage_block_shape_ab.txt / r06_stage_small_ab.txt set these shapes from the environment: a build with
    // -DLTP_EXP_KNOBS reads LTP_EXP_OF_JB, LTP_EXP_RS_JB, LTP_EXP_SS_PER; the product reads no environment)
#ifdef LTP_EXP_KNOBS
    static const int exp_of = getenv("LTP_EXP_OF_JB") ? atoi(getenv("LTP_EXP_OF_JB")) : 0, exp_rs = getenv("LTP_EXP_RS_JB") ? atoi(getenv("LTP_EXP_RS_JB")) : 0;
    static const int exp_ss = getenv("LTP_EXP_SS_PER") ? atoi(getenv("LTP_EXP_SS_PER")) : 0;
#else
    constexpr int exp_of = 0, exp_rs = 0, exp_ss = 0;
#endif
    const int jb_libm = (variant & kPowLibm) && jb > 4 ? 4 : jb;
    const dim3 block_of(kQueriesPerBlock, exp_of > 0 && exp_of <= jb ? exp_of : jb_libm), block_rs(kQueriesPerBlock, exp_rs > 0 && exp_rs <= jb ? exp_rs : jb_libm);
    dispatch_variant(variant, [&](auto v) {
        constexpr int SEM = decltype(v)::value;
        hipLaunchKernelGGL(k_opt_fast<SEM>, grid, block_of, 0, s, n, dof, t_sample, goal_check, lim, in, out, lane_flags, qa);
        hipLaunchKernelGGL(k_opt_slow<SEM>, dim3((unsigned)a_blocks), dim3(64), 0, s, dof, t_sample, lim, in, out, lane_flags, qa);
        hipLaunchKernelGGL(k_reduce_scale<SEM>, grid, block_rs, 0, s, n, dof, t_sample, lim, in, out, lane_flags, qb);
        hipLaunchKernelGGL(k_scaling_slow<SEM>, dim3((unsigned)b_blocks), dim3(kQueriesPerBlock, 8), 0, s, dof, t_sample, lim, in, out, qb, exp_ss);
    });
}

void launch_offsets(hipStream_t s, long long n, int dof, double t_sample, Records rec,
                    unsigned long long* block_sums, unsigned long long* offsets, bool lens_ready, RowSpec rows)
{
    if (n <= 0) return;
    const long long nb = (n + kScanBlock - 1) / kScanBlock;
    if (lens_ready) hipLaunchKernelGGL(k_finalize_lens, dim3((unsigned)nb), dim3(256), 0, s, n, dof, rows, rec, block_sums);
    else hipLaunchKernelGGL(k_finalize, dim3((unsigned)nb), dim3(256), 0, s, n, dof, t_sample, rows, rec, block_sums);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(1024), 0, s, nb, block_sums);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, s, n, dof, rows, rec.traj_len, block_sums, offsets);
}

}  // namespace ltp
