// ltp_kernels.hip — hand-written CDNA4 (gfx950) kernels for the batched planner.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile). No fast-math:
// the inf/NaN flow of the reference is part of the contract.
#include "ltp_kernels.hpp"
#include "ltp_profile.hpp"

namespace ltp {

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// 16-byte store unit of an output row: 2 doubles or 4 floats
template <typename T> struct OutVec;
template <> struct OutVec<double> { typedef double2_t type; static constexpr int N = 2; };
template <> struct OutVec<float> { typedef float4_t type; static constexpr int N = 4; };

LTP_DEV JointLimits load_limits(const Limits& lim, int j)
{
    JointLimits L;
    L.q_min = lim.q_min[j];
    L.q_max = lim.q_max[j];
    L.v_max = lim.v_max[j];
    L.a_max = lim.a_max[j];
    L.j_max = lim.j_max[j];
    return L;
}

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
// (int)ceil(t[6]/Ts) + 1 of one joint (cc:718), or -1 if any of its switching times is not finite or the length does
// not fit an int (both DEFINED here: the reference converts out-of-range doubles to int, which is undefined)
LTP_DEV int joint_len(const double (&t)[7], double t_sample)
{
    bool finite = true;
#pragma unroll
    for (int k = 0; k < 7; ++k) finite = finite && dfinite(t[k]);
    const double len = dceil(t[6] / t_sample) + 1.0;
    return (finite && len < 2147483647.0) ? (int)len : -1;
}

constexpr int kLaneGoalOutside = 128; // lane_flags bit: q_goal outside [q_min, q_max] (only with the opt-in goal check)
constexpr int kLaneDeferred = 64;   // lane_flags bit: optSwitchTimes of this lane is still pending in queue A

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

LTP_DEV void store_opt_record(const Records& out, long long rj, const double (&t)[7], double dir, int mod)
{
#pragma unroll
    for (int k = 0; k < 7; ++k) out.t_opt[rj * 7 + k] = t[k];
    out.dir[rj] = dir;
    out.mod[rj] = (signed char)mod;
}

__global__ void __launch_bounds__(kQueriesPerBlock* kMaxJointSlots)
k_opt_fast(long long n, int dof, double t_sample, int goal_check, Limits lim, Queries in, Records out,
           signed char* __restrict__ lane_flags, Queue queue)
{
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
            int flags = check_inputs_joint(L, q0, v0, a0) ? 0 : kStatusInvalidInput;
            // NEW, opt-in: the reference never checks q_goal (cc:68-77), only the last sample (cc:59-61)
            if (goal_check && !(qg >= L.q_min && qg <= L.q_max)) flags |= kLaneGoalOutside;
            double t[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            double dir = 0.0;
            int mod = 0;
            const int rc = opt_switch_times<false>(L.a_max, L.j_max, t_sample, qg, q0, v0, a0, L.v_max, t, dir, mod);
            if (rc == kOptDefer) {
                defer = true;
                flags |= kLaneDeferred;
            } else {
                if (rc == kOptFalse) flags |= kStatusOptFailed;
                store_opt_record(out, rj, t, dir, mod);
            }
            lane_flags[rj] = (signed char)flags;
        }
        block_push(defer, (unsigned long long)rj, queue, s_cnt);
    }
}

__global__ void __launch_bounds__(64)
k_opt_slow(int dof, double t_sample, Limits lim, Queries in, Records out, signed char* __restrict__ lane_flags, Queue queue)
{
    unsigned long long cnt[kQueueShards];
    const unsigned long long count = queue_total(queue, cnt);
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
        const int rc = opt_switch_times<true>(L.a_max, L.j_max, t_sample, in.q_goal[ix], in.q_0[ix], in.v_0[ix], in.a_0[ix],
                                              L.v_max, t, dir, mod);
        store_opt_record(out, rj, t, dir, mod);
        int flags = lane_flags[rj] & ~kLaneDeferred;
        if (rc == kOptFalse) flags |= kStatusOptFailed;
        lane_flags[rj] = (signed char)flags;
    }
}

__global__ void __launch_bounds__(kQueriesPerBlock* kMaxJointSlots)
k_reduce_scale(long long n, int dof, double t_sample, Limits lim, Queries in, Records out,
               const signed char* __restrict__ lane_flags, Queue queue)
{
    __shared__ unsigned long long s_cnt[kMaxJointSlots + 1];
    __shared__ double s_t[kMaxJointSlots][kQueriesPerBlock];
    __shared__ int s_j[kMaxJointSlots][kQueriesPerBlock];
    __shared__ int s_f[kMaxJointSlots][kQueriesPerBlock];

    const int x = threadIdx.x, y = threadIdx.y, JB = blockDim.y;
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
    if (live && y == 0) {
        out.t_required[q] = t_required;
        out.slowest[q] = slowest;
        out.status[q] = flags;
    }

    // cc:43-55 with the closed-form candidates c1, c2 (cc:378-446)
    int my_len = 0, nonfinite = 0;
    for (int jb = 0; jb < dof; jb += JB) {   // same number of rounds in every wave: block_push() contains barriers
        const int j = jb + y;
        const bool active = live && j < dof;
        const JointLimits L = load_limits(lim, j < dof ? j : dof - 1);
        bool need_slow = false;
        int lane_len = 0;
        const long long rj = q * dof + j;
        if (active) {
            double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            double vd = L.v_max;
            int mod = 0;   // failed query: zero record, never sampled
            if (flags == 0) {
                mod = out.mod[rj];
                if (j != slowest) {
                    const long long ix = q * in.sq + (long long)j * in.sj;
                    const double qg = in.q_goal[ix], q0 = in.q_0[ix];
                    double v0 = in.v_0[ix], a0 = in.a_0[ix];
                    const double dir = out.dir[rj];
                    if (dir < 0.0) { v0 = -v0; a0 = -a0; }
                    vd = v_drive_candidate<1>(L.a_max, L.j_max, qg, q0, v0, a0, dir, t_required);
                    int acc = try_v_drive<false>(L.a_max, L.j_max, t_sample, qg, q0, v0, a0, dir, t_required, vd, ts, mod);
                    if (acc == kOptFalse) {
                        vd = v_drive_candidate<2>(L.a_max, L.j_max, qg, q0, v0, a0, dir, t_required);
                        acc = try_v_drive<false>(L.a_max, L.j_max, t_sample, qg, q0, v0, a0, dir, t_required, vd, ts, mod);
                    }
                    need_slow = (acc != kOptTrue);
                }
                if (!need_slow) {
                    // cc:50-55: no scaled solution (or the slowest joint) -> optimal times
                    double mx = ts[0];
#pragma unroll
                    for (int k = 1; k < 7; ++k) if (mx < ts[k]) mx = ts[k];
                    if (mx <= 0.0) {
#pragma unroll
                        for (int k = 0; k < 7; ++k) ts[k] = out.t_opt[rj * 7 + k];
                    }
                }
            }
            if (!need_slow) {
#pragma unroll
                for (int k = 0; k < 7; ++k) out.t_scaled[rj * 7 + k] = ts[k];
                out.v_drive[rj] = vd;
                out.mod[rj] = (signed char)mod;
                if (flags == 0) lane_len = joint_len(ts, t_sample);
            }
        }
        block_push(need_slow, (unsigned long long)rj, queue, s_cnt);
        if (lane_len < 0) nonfinite = 1;
        else my_len = lane_len > my_len ? lane_len : my_len;
    }
    // traj_len (cc:716-719) over the joints finished here; queue-B lanes add theirs with atomicMax later
    __syncthreads();
    s_j[y][x] = my_len;
    s_f[y][x] = nonfinite;
    __syncthreads();
    if (live && y == 0) {
        int len = 0, bad = 0;
        for (int yy = 0; yy < JB; ++yy) {
            len = s_j[yy][x] > len ? s_j[yy][x] : len;
            bad |= s_f[yy][x];
        }
        out.traj_len[q] = flags == 0 ? len : 0;
        if (bad) out.status[q] = flags | kStatusNonFinite;
    }
}

template <int C>
LTP_DEV bool scaling_case(const JointLimits& L, double t_sample, double qg, double q0, double v0, double a0, double dir,
                          double tr, double& vd, double (&ts)[7], int& mod)
{
    vd = v_drive_candidate<C>(L.a_max, L.j_max, qg, q0, v0, a0, dir, tr);
    return try_v_drive<true>(L.a_max, L.j_max, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod) == kOptTrue;
}

// cc:358-645 for one (query, joint): the eight candidates in the reference's order, then the reset
LTP_DEV bool time_scaling_full(const JointLimits& L, double t_sample, double qg, double q0, double v0, double a0, double dir,
                               double tr, double& vd, double (&ts)[7], int& mod, int& which)
{
    if (dir < 0.0) { v0 = -v0; a0 = -a0; }   // cc:372-375
    which = 1;
    if (scaling_case<1>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 2;
    if (scaling_case<2>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 3;
    if (scaling_case<3>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 4;
    if (scaling_case<4>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 5;
    if (scaling_case<5>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 6;
    if (scaling_case<6>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 7;
    if (scaling_case<7>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 8;
    if (scaling_case<8>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod)) return true;
    which = 0;   // cc:640-644
    mod = 0;
    zero7(ts);
    vd = L.v_max;
    return false;
}

// Queue B. Block = 64 queued (query, joint) items x 8 waves; wave c evaluates candidate c+1 for all 64 items, so
// the eight candidates of cc:378-638 (independent computations) run side by side and the kernel's latency is the
// slowest candidate (the degree-6 solve) instead of their sum. The reference's "first accepted in order" is then a
// lookup over eight flags in LDS.
__global__ void __launch_bounds__(kQueriesPerBlock * 8)
k_scaling_slow(int dof, double t_sample, Limits lim, Queries in, Records out, Queue queue)
{
    __shared__ int s_acc[8][kQueriesPerBlock];
    const int x = threadIdx.x;
    const int c = __builtin_amdgcn_readfirstlane(threadIdx.y);
    unsigned long long cnt[kQueueShards];
    const unsigned long long count = queue_total(queue, cnt);
    for (unsigned long long base = (unsigned long long)blockIdx.x * kQueriesPerBlock; base < count;
         base += (unsigned long long)gridDim.x * kQueriesPerBlock) {
        const unsigned long long it = base + x;
        const bool live = it < count;
        bool acc = false;
        double ts[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double vd = 0.0;
        int mod = 0, j = 0;
        long long rj = 0, q = 0;
        JointLimits L = {0.0, 0.0, 0.0, 0.0, 0.0};
        if (live) {
            rj = (long long)queue_item(queue, cnt, it);
            q = rj / dof;
            j = (int)(rj - q * dof);
            L = load_limits(lim, j);
            const long long ix = q * in.sq + (long long)j * in.sj;
            const double qg = in.q_goal[ix], q0 = in.q_0[ix];
            double v0 = in.v_0[ix], a0 = in.a_0[ix];
            const double dir = out.dir[rj], tr = out.t_required[q];
            if (dir < 0.0) { v0 = -v0; a0 = -a0; }   // cc:372-375
            switch (c) {
            case 0: acc = scaling_case<1>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
            case 1: acc = scaling_case<2>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
            case 2: acc = scaling_case<3>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
            case 3: acc = scaling_case<4>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
            case 4: acc = scaling_case<5>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
            case 5: acc = scaling_case<6>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
            case 6: acc = scaling_case<7>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
            default: acc = scaling_case<8>(L, t_sample, qg, q0, v0, a0, dir, tr, vd, ts, mod); break;
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
            if (winner || reset) {
                double mx = ts[0];
#pragma unroll
                for (int k = 1; k < 7; ++k) if (mx < ts[k]) mx = ts[k];
                if (mx <= 0.0) {   // cc:50-55
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
// Packed layout of plan p at out + offsets[p]: [array q,v,a,j][joint][row_stride] doubles,
// row_stride = round_up(traj_len, 16) so that every row starts 128-B aligned.
// ---------------------------------------------------------------------------------------
// Samples stored per row: every rows.stride-th sample (0, stride, 2*stride, ...), at most rows.max_samples of them.
// {0, 1} stores whole trajectories, which is the reference's behaviour.
LTP_HD int stored_len(int len, RowSpec rows)
{
    if (len <= 0) return 0;
    const int st = rows.stride > 1 ? rows.stride : 1;
    const int cnt = (len + st - 1) / st;
    return (rows.max_samples > 0 && cnt > rows.max_samples) ? rows.max_samples : cnt;
}

LTP_DEV unsigned long long plan_size(int len, int dof)
{
    if (len <= 0) return 0ull;
    const unsigned long long stride = ((unsigned long long)len + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    return 4ull * (unsigned long long)dof * stride;
}

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
            if (st == 0) {
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
            if (rec.status[q] != 0) { len = 0; rec.traj_len[q] = 0; }   // failed or non-finite: nothing to sample
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

// Inside one run, with m = 1-based position in the run, S1 = m(m+1)/2 and S2 = m(m+1)(m+2)/6:
//   q(m) = q0 + (q1*m + (q2*S1 + q3*S2))   {q_s, Ts*v_s, Ts*Ts*a_s, Ts*Ts*Ts*J}
//   v(m) = v0 + (v1*m + v2*S1)             {v_s, Ts*a_s, Ts*Ts*J}
//   a(m) = a0 + a1*m                       {a_s, Ts*J}
//   j(m) = J
// and the three snap rules of cc:815-829 only change coefficients, so evaluating a sample has no branches.
constexpr int kRunCoefs = 10;   // q0..q3, v0..v2, a0, a1, J
struct RunCoef {
    double c[kRunCoefs];
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
struct SegTable {
    int start[kSampleJointGroup][kMaxSegments + 1];
    double c[kSampleJointGroup][kMaxSegments][kRunCoefs];
    int nseg[kSampleJointGroup];
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
// to eight "+=" fractional corrections (cc:768-807), applied in the reference's order. s = sampled switch indices,
// Jp = jerk of the seven phases, corr = the nine possible correction terms, all in LDS.
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
    if (s2 >= s1) {
        if (i == s0 + 1) val = val + corr[0];
        if (s1 > 0 && i == s1) val = val + corr[1];
        if (i == s2 + 1) val = val + corr[2];
    } else {
        if (s1 > 0 && i == s1) val = val + corr[3];
    }
    if (s3 > 0 && i == s3) val = val + corr[4];
    if (s2 - s0 > 0) {
        if (i == s4 + 1) val = val + corr[5];
    } else {
        if (s4 > 0 && i == s4) val = val + corr[6];
    }
    if (s5 > 0 && i == s5) val = val + corr[7];
    if (i == s6 + 1) val = val + corr[8];
    return val;
}

// coefficients of a run that starts after state (a_s, v_s, q_s)
LTP_DEV RunCoef run_coef(int mode, double J, double a_s, double v_s, double q_s, double vsnap, double Ts)
{
    RunCoef r;
#pragma unroll
    for (int x = 0; x < kRunCoefs; ++x) r.c[x] = 0.0;
    const double tj = Ts * J;
    r.c[9] = J;
    if (!(mode & kModeTail)) { r.c[7] = a_s; r.c[8] = tj; }
    r.c[0] = q_s;
    if (mode & kModeVSnap) {
        r.c[4] = vsnap;
        r.c[1] = Ts * vsnap;
    } else if (!(mode & kModeTail)) {
        r.c[4] = v_s; r.c[5] = Ts * a_s; r.c[6] = Ts * tj;
        r.c[1] = Ts * v_s; r.c[2] = Ts * (Ts * a_s); r.c[3] = Ts * (Ts * tj);
    }
    return r;
}

// the four outputs at position m of a run; the streaming loop and the state propagation both use exactly this
LTP_DEV void run_eval(const double (&c)[kRunCoefs], int m, double& q, double& v, double& a, double& j)
{
    const double md = (double)m;
    const double s1 = 0.5 * (md * (md + 1.0));
    const double s2 = s1 * (md + 2.0) * (1.0 / 3.0);
    q = c[0] + (c[1] * md + (c[2] * s1 + c[3] * s2));
    v = c[4] + (c[5] * md + c[6] * s1);
    a = c[7] + c[8] * md;
    j = c[9];
}

template <bool STREAMING, typename V>
LTP_DEV void store16(V* dst, V val)
{
    if constexpr (STREAMING) __builtin_nontemporal_store(val, dst);
    else *dst = val;
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
struct ItemRegs {
    int len;
    unsigned long long off;
    double pa, pb;
};

// Issues the loads of an item (nothing here waits for them). p < 0: no item.
LTP_DEV ItemRegs fetch_item(long long p, int j0, int nj, int dof, const Limits& lim, const Queries& in, const Records& rec,
                            const unsigned long long* __restrict__ offsets)
{
    ItemRegs r;
    r.len = 0; r.off = 0ull; r.pa = 0.0; r.pb = 0.0;
    if (p < 0) return r;
    r.len = rec.traj_len[p];
    if (offsets) r.off = offsets[p];
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
        tab.w.corr[jl][3] = ft[0] * J0 + d20 * J2;                        // j[s1]     cc:781 (phase 2 absent)
        tab.w.corr[jl][4] = (1 - ft[3]) * J4;                             // j[s3]     cc:787
        tab.w.corr[jl][5] = ft[4] * J4;                                   // j[s4+1]   cc:793
        tab.w.corr[jl][6] = ft[4] * J4 + ft[0] * J0 + d20 * J2;           // j[s4]     cc:798 (phases 2, 3 absent)
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
        if (mine) tab.start[jl][pos] = cval;
        if (k == 0) { tab.start[jl][distinct] = len; tab.nseg[jl] = distinct; }
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[6] = wall_clock64(); }
    // (4) lane k < ns: mode and jerk of run k, and everything of the run's end-state update that does not depend
    //     on the state (parked in tab.c[.][k][0..5] until step (6) overwrites it with the coefficients)
    const int ns = jact ? tab.nseg[jl] : 0;
    if (k < ns) {
        const int b = tab.start[jl][k];
        const int* sj = tab.w.s[jl];
        const bool phase4 = sj[3] - sj[2] > 2;                                         // cc:813
        int mode = 0;
        if (b > sj[6]) mode |= kModeTail;
        if (phase4 && b >= sj[2] + 1 && b < sj[3] - 1) mode |= kModeVSnap;
        const double J = jerk_at(sj, tab.w.Jp[jl], tab.w.corr[jl], b);
        tab.w.runMode[jl][k] = mode;
        tab.w.runJ[jl][k] = J;
        const double md = (double)(tab.start[jl][k + 1] - b);     // samples in the run
        const double s1 = 0.5 * (md * (md + 1.0));
        const double tj = Ts * J;
        double* pre = tab.c[jl][k];
        pre[0] = md;
        pre[1] = s1;
        pre[2] = s1 * (md + 2.0) * (1.0 / 3.0);
        pre[3] = tj;
        pre[4] = Ts * tj;
        pre[5] = Ts * (Ts * tj);
    }
    wave_sync();
    if constexpr (PROBE) { if (threadIdx.x == 0) probe[7] = wall_clock64(); }
    // (5) lane 0 of the joint: the state before each run — the only serial part. Each step is run_eval(run_coef(..))
    //     at the run's last sample, i.e. exactly what the streaming loop will store there, with the state-independent
    //     factors taken from step (4).
    if (jact && k == 0) {
        const double vsnap = tab.w.misc[jl][1];
        double q = tab.w.misc[jl][2], v = tab.w.misc[jl][3], a = tab.w.misc[jl][4];   // state "before sample 0" (cc:810-812)
        // software-pipelined by hand: the state-independent factors of run m+1 are fetched from LDS while the
        // dependent chain of run m executes (the chain is ~5 binary64 operations, an LDS round trip is longer)
        const double* pre = tab.c[jl][0];
        double md = pre[0], s1 = pre[1], s2 = pre[2], p3 = pre[3], p4 = pre[4], p5 = pre[5];
        int mode = tab.w.runMode[jl][0];
        for (int m = 0; m < ns; ++m) {
            const int mn = m + 1 < ns ? m + 1 : m;
            const double* nx = tab.c[jl][mn];
            const double md_n = nx[0], s1_n = nx[1], s2_n = nx[2], p3_n = nx[3], p4_n = nx[4], p5_n = nx[5];
            const int mode_n = tab.w.runMode[jl][mn];
            tab.w.state[jl][m][0] = a; tab.w.state[jl][m][1] = v; tab.w.state[jl][m][2] = q;
            double qn, vn, an;
            if (mode & kModeVSnap) {
                vn = vsnap + (0.0 * md + 0.0 * s1);
                qn = q + ((Ts * vsnap) * md + (0.0 * s1 + 0.0 * s2));
            } else if (mode & kModeTail) {
                vn = 0.0 + (0.0 * md + 0.0 * s1);
                qn = q + (0.0 * md + (0.0 * s1 + 0.0 * s2));
            } else {
                vn = v + ((Ts * a) * md + p4 * s1);
                qn = q + ((Ts * v) * md + ((Ts * (Ts * a)) * s1 + p5 * s2));
            }
            an = (mode & kModeTail) ? 0.0 + 0.0 * md : a + p3 * md;
            q = qn; v = vn; a = an;
            md = md_n; s1 = s1_n; s2 = s2_n; p3 = p3_n; p4 = p4_n; p5 = p5_n; mode = mode_n;
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
        for (int x = 0; x < kRunCoefs; ++x) tab.c[jl][k][x] = rc.c[x];
    }
    __builtin_amdgcn_s_setprio(0);
}

// Streams the rows of one item (plan x joint group) from the run tables in LDS. Every thread of the block calls this.
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

    // Pass B, once per item: the slots that contain a run boundary. There are at most 19 per row, but in the
    // row-by-row loop below most 64-slot wave steps contain one, and a wave that has one would execute the per-sample
    // path for all its lanes. So lane k of joint slot jl (the mapping of the table build) evaluates the slot of
    // boundary k, if that slot really straddles it and boundary k-1 has not claimed the same slot, and parks the four
    // 16-byte results in LDS (in the space of the build scratch); the main loop picks them up, so that it still
    // writes every row as full contiguous wave stores (leaving holes for scattered 16-byte stores costs 13 % of the
    // float64 bandwidth).
    if constexpr (!DRY) {
        const int jl = threadIdx.x >> 5, k = threadIdx.x & 31;
        const int nruns = jl < nj ? tab.nseg[jl] : 0;
        if (k >= 1 && k < nruns) {
            const int* st = tab.start[jl];
            const int u = (st[k] + sstride - 1) / sstride;             // first stored sample at or after boundary k
            bool mine = (u % N) != 0 && u < N * nslots;
            if (mine && k > 1) {
                const int up = (st[k - 1] + sstride - 1) / sstride;
                if ((up % N) != 0 && up / N == u / N) mine = false;     // boundary k-1 owns this slot
            }
            if (mine) {
                const int i0 = u / N * N, t0 = i0 * sstride;
                int kh = k - 1;
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
                    run_eval(tab.c[jl][kh], i - ch + 1, x4[0], x4[1], x4[2], x4[3]);
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
    for (int jl2 = wave >> lw; jl2 < nj; jl2 += 4 >> lw) {
        T* const row = plan_base + (unsigned long long)(j0 + jl2) * stride;
        const int* st = tab.start[jl2];
        const int nruns = tab.nseg[jl2];
        // run cursor of this lane: samples [cur, nxt) belong to run kr (nxt = INT_MAX for the last run)
        int kr = 0, cur = 0, nxt = nruns > 1 ? st[1] : 0x7fffffff;
        for (int slot = ((wave & ((1 << lw) - 1)) << 6) + lane; slot < nslots; slot += 64 << lw) {
            const int i0 = N * slot;                  // first stored sample of this slot; it is sample i0*sstride of the trajectory
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
                if (t0 + (N - 1) * sstride >= nxt) {
                    // run boundary kr+1 lies inside the slot: pass B has left the finished values in LDS
#pragma unroll
                    for (int x = 0; x < 4; ++x) o[x] = *reinterpret_cast<const V*>(&tab.bnd[jl2][kr + 1][x]);
                } else {
                    double c[kRunCoefs];
#pragma unroll
                    for (int x = 0; x < kRunCoefs; ++x) c[x] = tab.c[jl2][kr][x];
#pragma unroll
                    for (int h = 0; h < N; ++h) {
                        const bool pad = i0 + h >= slen;
                        double x4[4];
                        run_eval(c, t0 + h * sstride - cur + 1, x4[0], x4[1], x4[2], x4[3]);
#pragma unroll
                        for (int x = 0; x < 4; ++x) o[x][h] = pad ? (T)0 : (T)x4[x];
                    }
                }
            }
#pragma unroll
            for (int x = 0; x < 4; ++x) store16<STREAMING>(reinterpret_cast<V*>(row + x * arr_stride + i0), o[x]);
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
         unsigned long long* __restrict__ stamps, int spread, RowSpec rows, unsigned long long* __restrict__ next_item)
{
    __shared__ SegTable tab;
    __shared__ unsigned long long s_item;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    const long long per = (count + spread - 1) / spread;
    const unsigned long long total = (unsigned long long)per * spread * ngroups;
    const unsigned long long off0 = offsets[first];

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
        return fetch_item(some ? first + local : -1, j0, nj, dof, lim, in, rec, offsets);
    };

    if (threadIdx.x == 0) s_item = atomicAdd(next_item, 1ull);
    __syncthreads();
    unsigned long long item = s_item;
    ItemRegs cur = fetch(item);
    __syncthreads();   // s_item may be rewritten
    while (item < total) {
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        const long long p = first + local;
        const bool lead = threadIdx.x == 0 && j0 == 0;
        // diagnostic only (stamps == nullptr in every product call): start / tables ready / end on the 100 MHz wall clock
        if (stamps && lead && local < count) stamps[3 * local] = wall_clock64();
        const int len = cur.len;                          // 0: hole, failed or non-finite query -> nothing to sample
        const unsigned long long rel = cur.off - off0;
        const int slen = stored_len(len, rows);           // samples actually stored per row
        const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
        bool ok = len > 0;
        if (ok && rel + 4ull * dof * stride > capacity) {
            if (lead) atomicOr(&rec.status[p], kStatusOverflow);
            ok = false;
        }
        unsigned long long drawn = 0ull;
        if (threadIdx.x == 0) drawn = atomicAdd(next_item, 1ull);      // returns while the tables are built
        if (ok) build_run_tables(tab, p, j0, nj, len, t_sample, lim, rec, cur.pa, cur.pb);
        if (threadIdx.x == 0) s_item = drawn;
        __syncthreads();                                               // tables complete, next item known
        const unsigned long long nitem = s_item;
        const ItemRegs nxt = fetch(nitem);                             // in flight while this item streams
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
    const double s1 = 0.5 * (md * (md + 1.0));
    const double s2 = s1 * (md + 2.0) * (1.0 / 3.0);
    return c[0] + (c[1] * md + (c[2] * s1 + c[3] * s2));   // the q line of run_eval
}

template <bool PROBE>
__global__ void __launch_bounds__(kSampleThreads, kSampleBlocksPerCU)
k_envelope(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec, int window,
           int n_windows, int lg, double* __restrict__ env, unsigned long long* __restrict__ next_item,
           unsigned long long* __restrict__ probe_buf /* diagnostic, PROBE only: 16 stamps per item */)
{
    __shared__ SegTable tab;
    __shared__ unsigned long long s_item;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    const unsigned long long total = (unsigned long long)count * ngroups;
    for (;;) {
        __syncthreads();
        unsigned long long t_top = 0ull;
        if constexpr (PROBE) t_top = wall_clock64();
        if (threadIdx.x == 0) s_item = atomicAdd(next_item, 1ull);
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
        double2_t* const dst = reinterpret_cast<double2_t*>(env) + ((unsigned long long)local * dof + j0) * n_windows;
        if (len <= 0) {
            const double nan = __builtin_nan("");
            for (int task = threadIdx.x; task < tasks; task += kSampleThreads) dst[task] = double2_t{nan, nan};
            continue;
        }
        if constexpr (PROBE) { if (threadIdx.x == 0) probe[2] = wall_clock64(); }
        const ItemRegs regs = fetch_item(p, j0, nj, dof, lim, in, rec, nullptr);
        build_run_tables<PROBE>(tab, p, j0, nj, len, t_sample, lim, rec, regs.pa, regs.pb, probe);
        __syncthreads();
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
                const int* st = tab.start[jl];
                const int nruns = tab.nseg[jl];
                const long long b = (long long)w * window;
                const bool past = b >= (long long)len;                            // past the end: the last sample only
                int i = past ? len - 1 + r : (int)b + r;
                const int e = (b + window < (long long)len) ? (int)(b + window) : len;
                int kr = 0, cur = 0, nxt = nruns > 1 ? st[1] : 0x7fffffff;
                // the four q coefficients of the current run stay in registers; they are re-read at a run boundary only
                double c4[4] = {tab.c[jl][0][0], tab.c[jl][0][1], tab.c[jl][0][2], tab.c[jl][0][3]};
                for (; i < e; i += g) {
                    if (nxt <= i) {
                        do {
                            ++kr;
                            cur = nxt;
                            nxt = kr + 1 < nruns ? st[kr + 1] : 0x7fffffff;
                        } while (nxt <= i);
#pragma unroll
                        for (int x = 0; x < 4; ++x) c4[x] = tab.c[jl][kr][x];
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
                const unsigned long long* __restrict__ offsets, const T* __restrict__ tile,
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
    if (slen <= 0) {   // plan was not sampled: carry its start state over unchanged
        const long long ix = p * in.sq + (long long)j * in.sj;
        q_0[dst] = in.q_0[ix];
        v_0[dst] = in.v_0[ix];
        a_0[dst] = in.a_0[ix];
        return;
    }
    int k = sample_index ? sample_index[local] : uniform_index;
    k = k < 0 ? 0 : (k >= slen ? slen - 1 : k);   // beyond the stored samples: the last stored state
    const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    const T* row = tile + (offsets[p] - offsets[first]) + (unsigned long long)j * stride + k;
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
template <class Visit>
LTP_DEV void for_each_run(const Limits& lim, const Records& rec, long long rj, int j, int len, double Ts, double& q, double& v,
                          double& a, Visit&& visit)
{
    int sw[7];                                                                        // sampled switch indices (cc:751-757)
    double fr[7], frts[7];
#pragma unroll
    for (int x = 0; x < 7; ++x) {
        const double tk = rec.t_scaled[rj * 7 + x];
        fr[x] = tk - Ts * dfloor(tk / Ts);                                            // cc:747
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
    const double corr[9] = {frts[0] * J0, (1 - frts[1]) * J2, frts[2] * J2, frts[0] * J0 + d20 * J2, (1 - frts[3]) * J4,
                            frts[4] * J4, frts[4] * J4 + frts[0] * J0 + d20 * J2, (1 - frts[5]) * J6, frts[6] * J6};
    // candidate cut points (slot 0 = index 0 starts the first run and is not needed here)
    constexpr int cut_base[kCutSlots] = {0, 0, 0, 0, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 5, 6, 6, 6};
    constexpr int cut_delta[kCutSlots] = {0, 0, 1, 2, 0, 1, 0, 1, 2, -1, 0, 1, 0, 1, 2, 0, 1, 0, 1, 2};
    int cand[kCutSlots];
#pragma unroll
    for (int c = 1; c < kCutSlots; ++c) cand[c] = sw[cut_base[c]] + cut_delta[c];
    const bool phase4 = sw[3] - sw[2] > 2;                                            // cc:813
    int b = 0;
    for (int run = 0; run < kMaxSegments && b < len; ++run) {
        int e = len;                                                                  // next cut point after b
#pragma unroll
        for (int c = 1; c < kCutSlots; ++c) e = (cand[c] > b && cand[c] < e) ? cand[c] : e;
        int mode = 0;
        if (b > sw[6]) mode |= kModeTail;
        if (phase4 && b >= sw[2] + 1 && b < sw[3] - 1) mode |= kModeVSnap;
        const RunCoef rc = run_coef(mode, jerk_at(sw, Jp, corr, b), a, v, q, vsnap, Ts);
        if (visit(b, e, rc)) return;
        double jj;
        run_eval(rc.c, e - b, q, v, a, jj);
        b = e;
    }
}

// Receding horizon without any sampled rows: the state (q, v, a) at trajectory sample k of every plan straight from
// the switching-time records. A caller that only needs the restart state pays neither the table build of a sampler
// item (~15 us of latency per plan) nor a byte of trajectory traffic. The result has the bits of the row element the
// sampler would have stored at k.
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
        for_each_run(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [&](int b, int e, const RunCoef& rc) {
            if (k >= e) return false;
            double jj;
            run_eval(rc.c, k + 1 - b, q, v, a, jj);
            return true;
        });
    }
    q_0[dst] = q;
    v_0[dst] = v;
    a_0[dst] = a;
}

// ---------------------------------------------------------------------------------------
// Synthetic queries (distribution of reference tests/randomConfiguration.m:14-34
// generalised to per-joint limits). Counter-based: value = f(seed, query, joint, field), so any
// shard of any batch can be generated independently and the host reproduces it bit for bit.
// ---------------------------------------------------------------------------------------
LTP_DEV double unit_random(unsigned long long seed, unsigned long long query, unsigned int joint, unsigned int field)
{
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (((query * 64ull + joint) * 4ull + field) + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    return (double)(z >> 11) * 0x1.0p-53;
}

__global__ void __launch_bounds__(256)
k_generate(long long n, int dof, Limits lim, unsigned long long seed, long long first_query,
           double* __restrict__ q_goal, double* __restrict__ q_0, double* __restrict__ v_0, double* __restrict__ a_0,
           long long sq, long long sj)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * dof) return;
    const long long q = idx / dof;
    const int j = (int)(idx - q * dof);
    const JointLimits L = load_limits(lim, j);
    const unsigned long long gq = (unsigned long long)(first_query + q);
    const double eps = 1e-6;
    const double u0 = unit_random(seed, gq, j, 0), u1 = unit_random(seed, gq, j, 1);
    const double u2 = unit_random(seed, gq, j, 2), u3 = unit_random(seed, gq, j, 3);
    const double q0 = L.q_min + u0 * (L.q_max - L.q_min);
    const double qg = L.q_min + u1 * (L.q_max - L.q_min);
    const double vm = L.v_max - eps;
    const double v0 = -vm + u2 * (2.0 * vm);
    double a_lb, a_ub;
    if (v0 >= 0.0) {
        a_lb = -(L.a_max - eps);
        a_ub = dmin(L.a_max - eps, dsqrt(2.0 * L.j_max * (L.v_max - v0)));
    } else {
        a_lb = dmax(-(L.a_max - eps), -dsqrt(2.0 * L.j_max * (L.v_max - dabs(v0))));
        a_ub = L.a_max;
    }
    const double a0 = a_lb + u3 * (a_ub - a_lb);
    const long long ix = q * sq + (long long)j * sj;
    q_goal[ix] = qg;
    q_0[ix] = q0;
    v_0[ix] = v0;
    a_0[ix] = a0;
}

// ---------------------------------------------------------------------------------------
// One-lane mirrors of the protected member functions (for the reference's KAT-style tests).
// ---------------------------------------------------------------------------------------
// LongTermPlanner::checkInputs (cc:68-77) for one query
__global__ void k_check_inputs(int dof, Limits lim, const double* q_0, const double* v_0, const double* a_0, int* ok)
{
    int good = 1;
    for (int j = 0; j < dof; ++j)
        if (!check_inputs_joint(load_limits(lim, j), q_0[j], v_0[j], a_0[j])) good = 0;
    *ok = good;
}

__global__ void k_single_opt_braking(int joint, double t_sample, Limits lim, double v_0, double a_0, double* out)
{
    const JointLimits L = load_limits(lim, joint);
    double r[7] = {out[0], out[1], out[2], out[3], out[4], out[5], out[6]};
    double q, dir;
    opt_braking(L.a_max, L.j_max, t_sample, v_0, a_0, q, r, dir);
#pragma unroll
    for (int k = 0; k < 7; ++k) out[k] = r[k];
    out[7] = q;
    out[8] = dir;
}

__global__ void k_single_opt_switch(int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0, double a_0,
                                    double v_drive, double* io)
{
    const JointLimits L = load_limits(lim, joint);
    double t[7] = {io[0], io[1], io[2], io[3], io[4], io[5], io[6]};
    double dir = 0.0;
    int mod = 0;
    const bool ok = opt_switch_times<true>(L.a_max, L.j_max, t_sample, q_goal, q_0, v_0, a_0, v_drive, t, dir, mod) == kOptTrue;
#pragma unroll
    for (int k = 0; k < 7; ++k) io[k] = t[k];
    io[7] = dir;
    io[8] = (double)mod;
    io[9] = ok ? 1.0 : 0.0;
}

__global__ void k_single_time_scaling(int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0, double a_0,
                                      double dir, double tr, double* io)
{
    const JointLimits L = load_limits(lim, joint);
    double ts[7] = {io[0], io[1], io[2], io[3], io[4], io[5], io[6]};
    double vd;
    int mod = 0, which = 0;
    const bool acc = time_scaling_full(L, t_sample, q_goal, q_0, v_0, a_0, dir, tr, vd, ts, mod, which);
#pragma unroll
    for (int k = 0; k < 7; ++k) io[k] = ts[k];
    io[7] = vd;
    io[8] = (double)mod;
    io[9] = acc ? 1.0 : 0.0;
    io[10] = (double)which;
}

// device arithmetic probes: tests compare these with the host's libm bit for bit
__global__ void k_math_probe(long long n, const double* x, const double* y, double* out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double a = x[i], b = y[i];
    double* o = out + i * 8;
    o[0] = a / b;
    o[1] = dsqrt(dabs(a));
    o[2] = pw3(a);
    o[3] = pw4(a);
    o[4] = pw6(a);
    o[5] = dfloor(a / b);
    o[6] = dceil(a / b);
    o[7] = a * b + a;
}

template <int N>
LTP_DEV double probe_root(const double* c)
{
    double p[N + 1];
#pragma unroll
    for (int i = 0; i <= N; ++i) p[i] = c[i];
    return smallest_positive_real_root<N>(p);
}

__global__ void k_roots_probe(long long n, int degree, const double* coef, double* root)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* c = coef + i * 7;
    double r;
    if (degree == 4) r = probe_root<4>(c);
    else if (degree == 5) r = probe_root<5>(c);
    else r = probe_root<6>(c);
    root[i] = r;
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
                         unsigned long long* counts /* [16], zeroed by the caller on the same stream */)
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
    hipLaunchKernelGGL(k_opt_fast, grid, block, 0, s, n, dof, t_sample, goal_check, lim, in, out, lane_flags, qa);
    hipLaunchKernelGGL(k_opt_slow, dim3((unsigned)a_blocks), dim3(64), 0, s, dof, t_sample, lim, in, out, lane_flags, qa);
    hipLaunchKernelGGL(k_reduce_scale, grid, block, 0, s, n, dof, t_sample, lim, in, out, lane_flags, qb);
    hipLaunchKernelGGL(k_scaling_slow, dim3((unsigned)b_blocks), dim3(kQueriesPerBlock, 8), 0, s, dof, t_sample, lim, in, out, qb);
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

// how many blocks of a persistent (work-queue) kernel the device holds at once: 0 = k_sample float64 rows,
// 1 = k_sample float32 rows, 2 = k_envelope
int sample_resident_blocks(int device, int which)
{
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    hipError_t e;
    if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample<true, false, float>, kSampleThreads, 0);
    else if (which == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_envelope<false>, kSampleThreads, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample<true, false, double>, kSampleThreads, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 4;
    return cus * per_cu;
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
#define LTP_SAMPLE_CASE(ST, DR, TY) hipLaunchKernelGGL((k_sample<ST, DR, TY>), grid, block, 0, s, first, count, dof, t_sample, lim, in, rec, offsets, (TY*)out, capacity, stamps, spread, rows, next_item)
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

void launch_envelope(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in,
                     Records rec, int window, int n_windows, double* env, unsigned long long* next_item, int resident_blocks,
                     unsigned long long* probe)
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
        hipLaunchKernelGGL(k_envelope<true>, dim3((unsigned)blocks), dim3(kSampleThreads), 0, s, first, count, dof, t_sample, lim, in,
                           rec, window, n_windows, lg, env, next_item, probe);
    else
        hipLaunchKernelGGL(k_envelope<false>, dim3((unsigned)blocks), dim3(kSampleThreads), 0, s, first, count, dof, t_sample, lim, in,
                           rec, window, n_windows, lg, env, next_item, probe);
}

void launch_replan_states(hipStream_t s, long long first, long long count, int dof, RowSpec rows, Queries in, Records rec,
                          const unsigned long long* offsets, const void* tile, bool f32, const int* sample_index, int uniform_index,
                          double* q_0, double* v_0, double* a_0, long long sq, long long sj)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
    if (f32)
        hipLaunchKernelGGL(k_replan_states<float>, grid, block, 0, s, first, count, dof, rows, in, rec, offsets,
                           (const float*)tile, sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
    else
        hipLaunchKernelGGL(k_replan_states<double>, grid, block, 0, s, first, count, dof, rows, in, rec, offsets,
                           (const double*)tile, sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
}

void launch_state_at(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in,
                     Records rec, const int* sample_index, int uniform_index, double* q_0, double* v_0, double* a_0,
                     long long sq, long long sj)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    hipLaunchKernelGGL(k_state_at, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, first, count, dof, t_sample, lim, in, rec,
                       sample_index, uniform_index, q_0, v_0, a_0, sq, sj);
}

void launch_generate(hipStream_t s, long long n, int dof, Limits lim, unsigned long long seed, long long first_query,
                     double* q_goal, double* q_0, double* v_0, double* a_0, long long sq, long long sj)
{
    if (n <= 0) return;
    const long long total = n * dof;
    hipLaunchKernelGGL(k_generate, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, n, dof, lim, seed, first_query,
                       q_goal, q_0, v_0, a_0, sq, sj);
}

void launch_check_inputs(hipStream_t s, int dof, Limits lim, const double* q_0, const double* v_0, const double* a_0, int* ok)
{
    hipLaunchKernelGGL(k_check_inputs, dim3(1), dim3(1), 0, s, dof, lim, q_0, v_0, a_0, ok);
}
void launch_single_opt_braking(hipStream_t s, int joint, double t_sample, Limits lim, double v_0, double a_0, double* out10)
{
    hipLaunchKernelGGL(k_single_opt_braking, dim3(1), dim3(1), 0, s, joint, t_sample, lim, v_0, a_0, out10);
}
void launch_single_opt_switch(hipStream_t s, int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0,
                              double a_0, double v_drive, double* io10)
{
    hipLaunchKernelGGL(k_single_opt_switch, dim3(1), dim3(1), 0, s, joint, t_sample, lim, q_goal, q_0, v_0, a_0, v_drive, io10);
}
void launch_single_time_scaling(hipStream_t s, int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0,
                                double a_0, double dir, double t_required, double* out11)
{
    hipLaunchKernelGGL(k_single_time_scaling, dim3(1), dim3(1), 0, s, joint, t_sample, lim, q_goal, q_0, v_0, a_0, dir,
                       t_required, out11);
}
void launch_math_probe(hipStream_t s, long long n, const double* x, const double* y, double* out)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_math_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, x, y, out);
}
void launch_roots_probe(hipStream_t s, long long n, int degree, const double* coef, double* root)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_roots_probe, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, n, degree, coef, root);
}

}  // namespace ltp
