// ltp_sampler_walk.hip — the sampler whose run tables never leave the compute unit (round 4), gfx950: capped rows (first-N samples,
// receding-horizon rows), float32 rows, sparse rows — every format in which a plan's rows have too few bytes to hide the fused
// sampler's per-plan table build (ltp_sampler.hip), and on request any format.
//
// The table pass (k_build_tables + k_sample_tab*) pays a round trip through HBM per plan — 3 KB of packed tables written, 3.4 KB of
// lines read back — and the reads cost the write stream more than their bytes: that mixed pattern tops out at 5.3-5.6 TB/s of total
// traffic against 7.07 TB/s for pure row writes (profiles/EXPERIMENTS.md E6.3). Here one persistent block of 6 waves (four such blocks per
// compute unit; 8 waves and three blocks until the compact slot shrank in round 5) keeps the tables in LDS:
//   * the BUILDER wave (the last one) walks the runs of up to 63 (plan, joint) lanes at a time — a batch of plans — with
//     for_each_run (ltp_runs.hpp: the register walk of k_build_tables / k_state_at), leaves per lane the state before each of the
//     first kWalkRuns runs that start inside the row cap in one of two LDS batch buffers, applies the end-limit check (cc:59-61) and
//     the capacity rule, and publishes the batch;
//   * the five STREAMING waves take the (plan, joint) slots of the published batch, expand a run's coefficients on the fly
//     (run_coef: six LDS reads and ~15 operations per run instead of ten reads of an expanded table) and store the rows —
//     they read LDS and issue stores only, as in k_sample_tab;
//   * buffers change hands through LDS flags (s_ready / s_consumed), no block barrier in the loop; a final "done" batch lets every
//     wave leave.
// No table traffic, no table launch; the only global reads are the 13 record words per (plan, joint) and a plan's length. A batch is
// normally COMPACT (9 7-DoF plans — the next LIVE ones of the queue item in hand, walk_plans_per_item: rejected plans take no lane
// of a walk — the first kWalkRuns = 8 runs per lane); when one of its lanes has more runs inside the cap —
// a few per million of random queries, up to 4 % of the plans (a third of the batches) in the later cycles of a receding-horizon
// loop, tools/wide_batch_fraction.py —
// the builder rebuilds the same plans as WIDE batches (3 plans, all 20 runs per lane; beyond 21 joints: 21 joints of one plan at a
// time) in the same buffers. LONG rows (no cap, or a cap beyond 1024 samples) are wide batches from the start and are streamed one
// row per wave pass (walk_stream_rows). Rows are bit-identical
// to every other sampler's: same run walk, same run_coef / run_eval (include/ltp_run_tables.hpp).
// Both semantics: the builder's walk is for_each_run<SEM> (ltp_runs.hpp), the streaming side does not depend on it.
#include "ltp_sampler_lds.hpp"

#include <type_traits>

namespace ltp {

constexpr int kWalkRuns = 8;                                  // runs per (plan, joint) of a COMPACT batch
constexpr int kWalkLanes = 63;                                // (plan, joint) lanes of a compact batch: 9 plans of 7 joints
constexpr int kWalkMaxPlans = 9;
constexpr int kWideLanes = 21;                                // lanes of a WIDE batch (all kMaxSegments runs per lane): 3 plans of 7 joints
constexpr int kWalkStreamWaves = 5;
constexpr int kWalkThreads = (kWalkStreamWaves + 1) * 64;
constexpr int kWalkBuffers = 2;
// COMPACT slot, 76 words: a run is four doubles (a, v, q before the run, its jerk); its mode bits (kMode*, 3 of them) ride in the
// top four bits of its start sample, so every stored start — and the cap times the sample stride — must stay below 2^28
// (kWalkCompactEnd; rows beyond that are built wide). Round 5: 384 -> 304 bytes per lane, i.e. two batch buffers in 38.5 KB and FOUR
// blocks — four builder waves, one per SIMD — on a compute unit instead of three (profiles/EXPERIMENTS.md E7.9).
constexpr unsigned kWalkStartMask = 0x0fffffffu;
constexpr long long kWalkCompactEnd = 0x0fffffffll;
struct WalkSlot {
    static constexpr int kRuns = kWalkRuns;
    int nseg;                                                 // runs stored (<= kRuns)
    unsigned start[kWalkRuns + 1];                            // low 28 bits: first sample of run r; start[nseg]: first sample NOT covered (>= the cap, or traj_len); top 4 bits: run r's mode
    double vsnap;                                             // v_drive * dir (cc:823)
    double run[kWalkRuns][4];                                 // a, v, q before the run, its jerk
    LTP_DEV int first(int r) const { return (int)(start[r] & kWalkStartMask); }
    LTP_DEV int mode(int r) const { return (int)(start[r] >> 28); }
    LTP_DEV void put(int r, int b, int m, double a, double v, double q, double J)
    {
        start[r] = (unsigned)b | ((unsigned)m << 28);
        run[r][0] = a; run[r][1] = v; run[r][2] = q; run[r][3] = J;
    }
    LTP_DEV void close(int runs, int last_b) { start[runs] = (unsigned)(last_b < (int)kWalkStartMask ? last_b : (int)kWalkStartMask); nseg = runs; }
};
// WIDE slot, 114 words: every run of the lane, the layout of the packed tables (mode as a fifth word; starts use all 31 bits)
struct WideSlot {
    static constexpr int kRuns = kMaxSegments;
    int nseg;
    int pad0;
    int start[kMaxSegments + 2];
    double vsnap;
    double pad1;
    double run[kMaxSegments][kPackedRunWords];
    LTP_DEV int first(int r) const { return start[r]; }
    LTP_DEV int mode(int r) const { return (int)(unsigned)__builtin_bit_cast(unsigned long long, run[r][4]); }
    LTP_DEV void put(int r, int b, int m, double a, double v, double q, double J)
    {
        start[r] = b;
        run[r][0] = a; run[r][1] = v; run[r][2] = q; run[r][3] = J;
        run[r][4] = __builtin_bit_cast(double, (unsigned long long)(unsigned)m);
    }
    LTP_DEV void close(int runs, int last_b) { start[runs] = last_b; nseg = runs; }
};
static_assert(sizeof(WalkSlot) == 304 && sizeof(WideSlot) == 912, "76 / 114 words");
static_assert(kWalkLanes * sizeof(WalkSlot) == kWideLanes * sizeof(WideSlot), "both kinds of batch fill the buffer");
struct WalkBatch {
    union {
        WalkSlot slot[kWalkLanes];                            // compact batch: up to 9 plans, 8 runs per lane
        WideSlot wslot[kWideLanes];                           // wide batch: up to 3 plans, every run of every lane
    };
    unsigned long long rel0;                                  // element offset in `out` of the first row of the batch's first sampled plan
    unsigned long long span;                                  // elements from rel0 to the end of the batch's last sampled plan
    int nplans;                                               // plans in the batch
    int done;                                                 // 1 = the queue is exhausted
    int wide;                                                 // which member of the union holds the batch
    int j0, nj;                                               // the batch holds joints [j0, j0 + nj) of each of its plans (slots = nplans * nj):
                                                              // all of them up to 63 (compact) / 21 (wide) joints, a part of ONE plan beyond
    int pad;
    int slen[kWalkMaxPlans];                                  // stored samples per row of plan k of the batch; 0 = nothing to stream
    unsigned rel[kWalkMaxPlans];                              // row offset of plan k relative to rel0, in units of kRowAlign elements
};
static_assert(kWalkBuffers * sizeof(WalkBatch) + 64 <= 40 * 1024, "four blocks per compute unit (160 KB of LDS)");

// rows this kernel takes: every format, any number of joints (a compact batch holds whole plans up to 63 joints and 63 joints of
// one plan at a time beyond that; a wide batch whole plans up to 28 joints, 28 joints of one plan at a time beyond)
bool sample_walk_applies(int dof, RowSpec rows)
{
    if (dof < 1 || rows.max_samples < 0) return false;
    // capped rows up to kWalkBatchCap samples go through walk_stream, whose offsets inside a batch are 32-bit BYTE offsets behind one
    // buffer descriptor: the four arrays of a plan (4 * dof * row stride elements of at most 8 bytes) must stay below 2 GiB (round-4
    // advisor). That holds up to dof ~ 65 000 at a 1024-sample cap; beyond, the fused sampler / the table pass take the rows.
    if (rows.max_samples > 0 && rows.max_samples <= 1024) {
        const unsigned long long stride = ((unsigned long long)rows.max_samples + (kRowAlign - 1)) / kRowAlign * kRowAlign;
        if (4ull * (unsigned long long)dof * stride * 8ull >= (1ull << 31)) return false;
    }
    return true;
}

// LONG rows — no cap, or a cap beyond kWalkBatchCap samples: wide batches only, one row per wave pass (walk_stream_rows). Short rows: compact
// batches, several rows per pass, one descriptor with 32-bit offsets over the batch (walk_stream).
constexpr int kWalkBatchCap = 1024;
__host__ __device__ inline bool walk_long_rows(RowSpec rows) { return rows.max_samples <= 0 || rows.max_samples > kWalkBatchCap; }

// plans per batch: a compact batch; for long rows two wide batches
__host__ __device__ inline int walk_plans_per_batch(int dof, RowSpec rows)
{
    if (dof > kWalkLanes) return 1;                                                // one plan, kWalkLanes joints at a time
    const int compact = (kWalkLanes / dof) < kWalkMaxPlans ? (kWalkLanes / dof) : kWalkMaxPlans;
    if (!walk_long_rows(rows)) return compact;
    const int two_wide = 2 * (kWideLanes / dof) > 1 ? 2 * (kWideLanes / dof) : 1;
    return two_wide < compact ? two_wide : compact;
}
// plans per QUEUE ITEM. Rows of at most kWalkGatherCap samples — where the builder's walks are what a block waits for — take
// kWalkGather batches' worth of consecutive plans per item, and a batch is made of the item's LIVE plans — the ones that store
// samples — walk_plans_per_batch at a time: plans that were rejected (traj_len 0) have no rows and take no lane of a walk. (Round 5:
// in the later cycles of a receding-horizon loop a third of the random plans are dead; a batch of nine consecutive plans then walked
// six. Rows of the live plans of an item are neighbours in the tile whatever lies between them. Longer rows are bound by their
// stores: an item stays one batch there — gathered items cost first-256 1.3 % on one box, profiles/EXPERIMENTS.md E7.7.)
constexpr int kWalkGather = 3;
constexpr int kWalkGatherCap = 64;
__host__ __device__ inline int walk_plans_per_item(int dof, RowSpec rows)
{
    const int ppb = walk_plans_per_batch(dof, rows);
    if (rows.max_samples <= 0 || rows.max_samples > kWalkGatherCap) return ppb;
    const int g = 64 / ppb < kWalkGather ? (64 / ppb > 1 ? 64 / ppb : 1) : kWalkGather;     // (one traj_len load per lane)
    return g * ppb;
}
// The work queue of a launch: items of walk_plans_per_item consecutive plans, interleaved over `spread` stripes of the call's plans
// (item -> stripe item % spread, place item / spread; holes included).
struct WalkQueue {
    int ipp, ppb, spread;
    long long count, items, per;
    unsigned long long total;
};
__host__ __device__ inline WalkQueue walk_queue(long long count, int dof, RowSpec rows, int spread)
{
    WalkQueue q;
    q.ppb = walk_plans_per_batch(dof, rows);
    q.ipp = walk_plans_per_item(dof, rows);
    q.spread = spread > 0 ? spread : 1;
    q.count = count;
    q.items = (count + q.ipp - 1) / q.ipp;
    q.per = (q.items + q.spread - 1) / q.spread;
    q.total = (unsigned long long)q.per * (unsigned long long)q.spread;
    return q;
}
// first plan (local number) and plan count of a queue item; 0 plans: a hole of the interleave, or the end of the queue
__host__ __device__ inline void walk_queue_item(const WalkQueue& q, unsigned long long item, long long& pb, int& np)
{
    pb = 0;
    np = 0;
    if (item >= q.total) return;
    const long long bi = (long long)(item % (unsigned long long)q.spread) * q.per + (long long)(item / (unsigned long long)q.spread);
    if (bi < q.items) {
        pb = bi * q.ipp;
        np = (int)(q.count - pb < q.ipp ? q.count - pb : q.ipp);
    }
}

LTP_DEV unsigned long long walk_uniform(unsigned long long x)     // a value every lane holds alike -> scalar registers
{
    return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32)) << 32) |
           (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
}

// a streaming lane's place in its row's runs
struct WalkCursor {
    int kr, cur, nxt;                                         // run in use, its first sample, the first sample of the next run
};

// The pair of samples of one slot (stored samples i0, i0 + 1 of a row of slen; trajectory samples i * sstride) from the runs of W.
// A slot is a PAIR for float32 rows too (8-byte stores, a wave instruction still writes 512 contiguous bytes): four samples per lane
// in flight need 123 VGPRs, i.e. 2 blocks per CU, and lost to this form at every cap (profiles/EXPERIMENTS.md E6.7).
template <typename T, class Slot>
LTP_DEV void walk_eval_slot(const Slot& W, int nruns, WalkCursor& c, int i0, int sstride, int slen, double Ts, T __attribute__((ext_vector_type(2))) (&o)[4])
{
    constexpr int RUNS = Slot::kRuns;
    const int t0 = i0 * sstride;
    while (c.nxt <= t0) {
        ++c.kr;
        c.cur = c.nxt;
        c.nxt = c.kr + 1 < nruns ? W.first(c.kr + 1) : 0x7fffffff;
    }
    // first sample: the cursor's run
    int kh = c.kr, ch = c.cur, nh = c.nxt;
    RunCoef rc = run_coef<kSemMatlab>(W.mode(kh), W.run[kh][3], W.run[kh][0], W.run[kh][1], W.run[kh][2], W.vsnap, Ts);
    double x0[4], x1[4];
    run_eval(rc.c, t0 - ch + 1, x0[0], x0[1], x0[2], x0[3]);
    // second sample: the same run unless a boundary lies between the two (its coefficients then replace the first one's)
    const int i1 = t0 + sstride;
    if (nh <= i1) {
        do {
            ++kh;
            ch = nh;
            nh = kh + 1 < nruns ? W.first(kh + 1) : 0x7fffffff;
        } while (nh <= i1);
        const int kk = kh < RUNS ? kh : RUNS - 1;
        rc = run_coef<kSemMatlab>(W.mode(kk), W.run[kk][3], W.run[kk][0], W.run[kk][1], W.run[kk][2], W.vsnap, Ts);
    }
    run_eval(rc.c, i1 - ch + 1, x1[0], x1[1], x1[2], x1[3]);
    // (the tail of the last slot is row padding: zero)
    const bool pad0 = i0 >= slen, pad1 = i0 + 1 >= slen;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        o[x][0] = pad0 ? (T)0 : (T)x0[x];
        o[x][1] = pad1 ? (T)0 : (T)x1[x];
    }
}

template <bool STREAMING, typename V>
LTP_DEV void walk_store(V o, __amdgpu_buffer_rsrc_t rsrc, unsigned voff)
{
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (sizeof(V) == 16) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsrc, voff, 0, STREAMING ? /*nt | sc1*/ (2 | 16) : 0);
    else __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rsrc, voff, 0, STREAMING ? /*nt | sc1*/ (2 | 16) : 0);
}

// lanes per row (as a power of two) of a capped row format: the cap bounds every row of the call
__host__ __device__ inline int walk_row_lanes_log2(RowSpec rows)
{
    const int max_slots = (rows.max_samples + 1) / 2;
    // (round 5: down to one lane per row — at a cap of 16 samples the old floor of 16 lanes per row left half of every pass idle)
    return max_slots > 32 ? 6 : (max_slots > 16 ? 5 : (max_slots > 8 ? 4 : (max_slots > 4 ? 3 : (max_slots > 2 ? 2 : (max_slots > 1 ? 1 : 0)))));
}

// CAPPED rows: several rows per wave pass when they are short, every row of the batch behind one descriptor
template <bool STREAMING, typename T, class Slot>
LTP_DEV void walk_stream(const WalkBatch& B, const Slot* __restrict__ slots, int dof, T* __restrict__ out, RowSpec rows, double Ts, int wave,
                         int stream_waves = kWalkStreamWaves /* waves that share the batch's rows; 1: the calling wave writes them all */)
{
    constexpr int N = 2;
    typedef T V __attribute__((ext_vector_type(N)));                                  // what a lane stores per array and slot: 16 or 8 bytes
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int sstride = rows.stride > 1 ? rows.stride : 1;
    const int nplans = __builtin_amdgcn_readfirstlane(B.nplans);
    const int nj = __builtin_amdgcn_readfirstlane(B.nj), j0 = __builtin_amdgcn_readfirstlane(B.j0);
    const int total = nplans * nj;
    // lanes per row: the cap bounds every row of the call (wave-uniform, the same in every batch)
    const int lg = walk_row_lanes_log2(rows);
    const int rows_per_pass = 64 >> lg;
    // s / nj for s < 64, nj <= 63 without the integer-division sequence: (s + 0.5) / nj is at least 0.5 / 63 away from every integer
    const float inv_nj = 1.0f / (float)nj;
    // one buffer descriptor over the batch's rows (they are neighbours in the tile; at most 63 rows x 4 arrays of <= 1024 samples)
    // (rel0 and span are the same for the whole wave: made scalar, or every store gets a loop that checks its descriptor for uniformity)
    const unsigned long long rel0 = walk_uniform(B.rel0), span = walk_uniform(B.span);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(out + rel0, 0, (int)(unsigned)(span * sizeof(T)), 0x00020000);
    for (int s0 = wave * rows_per_pass; s0 < total; s0 += stream_waves * rows_per_pass) {
        const int s = s0 + (lane >> lg);                                          // this lane's (plan, joint) slot
        const bool in = s < total;
        const int pl = in ? (int)(((float)s + 0.5f) * inv_nj) : 0, j = in ? j0 + (s - pl * nj) : 0;
        const int slen = in ? B.slen[pl] : 0;
        if (__builtin_amdgcn_ballot_w64(slen > 0) == 0ull) continue;
        const unsigned stride = ((unsigned)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
        const unsigned arr_bytes = (unsigned)dof * stride * (unsigned)sizeof(T);
        const unsigned row_bytes = (B.rel[pl] * (unsigned)kRowAlign + (unsigned)j * stride) * (unsigned)sizeof(T);   // q row, bytes from the descriptor base
        constexpr int NF = OutVec<T>::N;                                          // the other samplers' slot: rows are zero-padded to its end
        const int nslots = slen > 0 ? (slen + NF - 1) / NF * (NF / N) : 0;
        const Slot& W = slots[in ? s : 0];
        const int nruns = W.nseg;
        WalkCursor c = {0, 0, nruns > 1 ? W.first(1) : 0x7fffffff};
        for (int slot = lane & ((1 << lg) - 1); slot < nslots; slot += 1 << lg) {
            V o[4];
            walk_eval_slot<T, Slot>(W, nruns, c, N * slot, sstride, slen, Ts, o);
            const unsigned voff = row_bytes + (unsigned)(N * slot) * (unsigned)sizeof(T);
#pragma unroll
            for (int x = 0; x < 4; ++x) walk_store<STREAMING>(o[x], rsrc, voff + (unsigned)x * arr_bytes);
        }
    }
}

// LONG rows (no cap, or a cap beyond kWalkBatchCap): wide batches only, one (plan, joint) row per wave pass, per row and array a
// descriptor over a window of the row (as stream_rows of the fused sampler: rows may be longer than 32-bit offsets reach)
constexpr int kWalkWindowSlots = 1 << 24;
template <bool STREAMING, typename T>
LTP_DEV void walk_stream_rows(const WalkBatch& B, int dof, T* __restrict__ out, RowSpec rows, double Ts, int wave)
{
    constexpr int N = 2;
    typedef T V __attribute__((ext_vector_type(N)));
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int sstride = rows.stride > 1 ? rows.stride : 1;
    const int nplans = __builtin_amdgcn_readfirstlane(B.nplans);
    const int nj = __builtin_amdgcn_readfirstlane(B.nj), j0 = __builtin_amdgcn_readfirstlane(B.j0);
    const int total = nplans * nj;
    const unsigned long long rel0 = walk_uniform(B.rel0);
    for (int s = wave; s < total; s += kWalkStreamWaves) {                        // (wave-uniform: scalar arithmetic below)
        const int pl = s / nj, j = j0 + (s - pl * nj);
        const int slen = __builtin_amdgcn_readfirstlane(B.slen[pl]);
        if (slen <= 0) continue;
        const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
        const unsigned long long arr = (unsigned long long)dof * stride;
        T* row = out + rel0 + (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)B.rel[pl]) * kRowAlign + (unsigned long long)j * stride;
        constexpr int NF = OutVec<T>::N;
        const int nslots = (slen + NF - 1) / NF * (NF / N);
        const WideSlot& W = B.wslot[s];
        const int nruns = __builtin_amdgcn_readfirstlane(W.nseg);
        WalkCursor c = {0, 0, nruns > 1 ? W.first(1) : 0x7fffffff};
        for (int wbase = 0; wbase < nslots; wbase += kWalkWindowSlots) {
            const int wend = nslots - wbase < kWalkWindowSlots ? nslots : wbase + kWalkWindowSlots;
            __amdgpu_buffer_rsrc_t rsrc[4];
#pragma unroll
            for (int x = 0; x < 4; ++x)
                rsrc[x] = __builtin_amdgcn_make_buffer_rsrc(row + x * arr + (unsigned long long)wbase * N, 0, (wend - wbase) * (int)sizeof(V), 0x00020000);
            for (int slot = wbase + lane; slot < wend; slot += 64) {
                V o[4];
                walk_eval_slot<T, WideSlot>(W, nruns, c, N * slot, sstride, slen, Ts, o);
                const unsigned voff = (unsigned)(slot - wbase) * (unsigned)sizeof(V);
#pragma unroll
                for (int x = 0; x < 4; ++x) walk_store<STREAMING>(o[x], rsrc[x], voff);
            }
        }
    }
}

// what the builder holds of its (plan, joint) lane between "loads issued" and "walk"
struct WalkLaneIn {
    JointRecord R;
    double q0, v0, a0, j_max, q_min, q_max;
    unsigned long long rel;
    int len;
};

// The walk of one lane into its slot of the batch under construction. Returns true if the lane has more runs inside the cap than
// the slot holds (compact batches only: the batch is then rebuilt wide). q_end receives the last trajectory sample (cc:59-61) —
// unless stop_at_cap: the walk then ends at the first run that is not needed (no end-limit verdict: LTPlanner.m has none, and a
// caller of the C++ semantics may ask for rows without it, ltp_sample_batch flags bit 4).
template <int SEM, bool LEAN, class Slot>
LTP_DEV bool walk_lane(Slot& W, const WalkLaneIn& L, long long needed_end, bool stop_at_cap, double Ts, double& q_end, bool last_joint)
{
    constexpr int RUNS = Slot::kRuns;
    double q = L.q0, v = L.v0, a = L.a0;
    int runs = 0, last_b = L.len;
    bool too_many = false;
    for_each_run_loaded<SEM, LEAN>(L.R, L.j_max, L.len, Ts, q, v, a, [&](int b, int, const RunCoef& rc) {
        if ((long long)b < needed_end) {
            if (runs < RUNS) {
                // q, v, a still hold the state before this run: the walk advances them after the visit
                W.put(runs, b, rc.mode, a, v, q, rc.c[9]);
                ++runs;
            } else {
                too_many = true;
            }
        } else if (last_b == L.len) {
            last_b = b;                                                          // first run that is not needed: it ends the last stored one
        }
        // otherwise the walk goes on for the end-limit check — to the first tail run (C++ semantics: a = v = 0 there, q rests)
        return (long long)b >= needed_end && (stop_at_cap || (SEM != kSemMatlab && (rc.mode & kModeTail) != 0));
    }, last_joint);
    W.close(runs, last_b);
    W.vsnap = L.R.v_drive * L.R.dir;                                             // as the walk forms it (cc:823)
    q_end = q;
    return too_many;
}

// everything a wave needs to build batches of one launch
struct WalkCtx {
    long long first, count;
    int dof;
    double t_sample;
    Limits lim;
    Queries in;
    Records rec;
    const unsigned long long* offsets;
    unsigned long long off0, capacity;
    RowSpec rows;
    long long needed_end;                                      // runs that start at or after this trajectory sample are not needed
    int lane;
};

// The trajectory lengths of an item's plans: lane k < np_item loads plan pb + k's (issued one item ahead by the callers: nothing waits)
LTP_DEV int walk_item_len_load(const WalkCtx& c, long long pb, int np_item)
{
    return c.lane < np_item ? c.rec.traj_len[c.first + pb + c.lane] : 0;
}
// The live plans of one queue item (np_item consecutive plans; len: what walk_item_len_load returned for it). Returns in lane r the
// item-relative index of the r-th live plan; nlive: their number.
LTP_DEV int walk_item_plans(const WalkCtx& c, int len, int np_item, int& nlive)
{
    const bool live = c.lane < np_item && stored_len(len, c.rows) > 0;
    const unsigned long long mask = walk_uniform(__builtin_amdgcn_ballot_w64(live));
    nlive = __builtin_popcountll(mask);
    if (nlive == np_item) return c.lane;                                           // every plan is live (wave-uniform)
    // lane i of a live plan knows its rank (live plans below it); the inverse — rank -> lane — is one forward permute. Dead lanes
    // push to lane 63, which is no rank here (nlive < np_item <= 64).
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
    return __builtin_amdgcn_ds_permute((live ? rank : 63) << 2, c.lane);
}

// Builds one batch — joints [j0, j0 + nj) of np live plans of an item, entries [base, base + np) of its plan list (walk_item_plans;
// local plan = pb + entry) — into B: record loads, the walk of every lane into its slot
// (compact: the first kWalkRuns runs inside the cap; WIDE: every run), the end-limit verdict (cc:59-61) and the tile-capacity rule, the
// plan-level header. The calling wave owns B. Returns false without a valid header if a compact batch does not do: a lane has more
// runs inside the cap than a slot holds, or a plan lies wholly inside the cap (nearly always more than kWalkRuns runs: wide at once).
#ifndef LTP_WALK_BUILDER_LEAN
#define LTP_WALK_BUILDER_LEAN 0
#endif
constexpr bool kWalkBuilderLean = LTP_WALK_BUILDER_LEAN != 0;
#ifndef LTP_WALK_AUTO_LEAN
#define LTP_WALK_AUTO_LEAN 1
#endif
constexpr bool kWalkAutoLean = LTP_WALK_AUTO_LEAN != 0;         // the autonomous waves (every wave walks AND streams): the branch-free form   // the builder wave of the builder / streaming-wave form: see for_each_run_loaded
template <int SEM, bool WIDE, bool STOP, bool LEAN>
LTP_DEV bool walk_build(const WalkCtx& c, WalkBatch& B, long long pb, int plist, int base, int np, int j0, int nj)
{
    const int lane = c.lane;
    // the loads of lane (batch plan lane / nj, joint j0 + lane % nj); nothing here waits
    WalkLaneIn L;
    const int pl = lane / nj, jl = lane - pl * nj;
    const bool mine = pl < np;
    const long long pmine = c.first + pb + __shfl(plist, base + (mine ? pl : 0));   // this lane's plan
    L.len = 0;
    if (mine) {
        const int j = j0 + jl;
        const long long p = pmine;
        const long long ix = p * c.in.sq + (long long)j * c.in.sj;
        L.len = c.rec.traj_len[p];
        L.rel = c.offsets[p] - c.off0;
        L.R = load_joint_record(c.rec, p * c.dof + j);
        L.q0 = c.in.q_0[ix]; L.v0 = c.in.v_0[ix]; L.a0 = c.in.a_0[ix];
        L.j_max = c.lim.j_max[j]; L.q_min = c.lim.q_min[j]; L.q_max = c.lim.q_max[j];
    }
    if constexpr (!WIDE) {
        // (the lengths arrive with the rest of the records: no round trip of their own)
        if (__builtin_amdgcn_ballot_w64(L.len > 0 && (long long)L.len <= c.needed_end) != 0ull) return false;
    }
    const long long p = pmine;
    int slen = mine ? stored_len(L.len, c.rows) : 0;
    const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    if (slen > 0 && L.rel + 4ull * c.dof * stride > c.capacity) {
        if (jl == 0) atomicOr(&c.rec.status[p], kStatusOverflow);
        slen = 0;
    }
    bool too_many = false;
    if (slen > 0) {
        double q_end;
        if constexpr (WIDE) too_many = walk_lane<SEM, LEAN>(B.wslot[lane], L, c.needed_end, STOP, c.t_sample, q_end, j0 + jl == c.dof - 1);
        else too_many = walk_lane<SEM, LEAN>(B.slot[lane], L, c.needed_end, STOP, c.t_sample, q_end, j0 + jl == c.dof - 1);
        if constexpr (SEM == kSemCpp && !STOP) {                                             // (LTPlanner.m has no position limits)
            if (q_end < L.q_min || q_end > L.q_max) atomicOr(&c.rec.status[p], kStatusEndLimit);   // cc:59-61: the last sample
        }
    }
    if (__builtin_amdgcn_ballot_w64(too_many) != 0ull) return false;
    // plan-level header: lane (plan pl, first joint of the batch) holds the plan's stored length and row offset
    if (lane < kWalkMaxPlans) { B.slen[lane] = 0; B.rel[lane] = 0u; }
    wave_sync();
    if (mine && jl == 0) B.slen[pl] = slen;
    wave_sync();
    // the span of rows this batch writes: from the first sampled plan to the end of the last one (plans are neighbours in the tile)
    const int sl = lane < np ? B.slen[lane] : 0;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(sl > 0);
    const int src = lane < np ? lane * nj : 0;                                                // lane k < np takes plan k's row offset from the plan's first lane
    const unsigned long long my_rel = ((unsigned long long)(unsigned)__shfl((int)(unsigned)(L.rel >> 32), src) << 32) |
                                      (unsigned long long)(unsigned)__shfl((int)(unsigned)L.rel, src);
    unsigned long long r_lo = 0ull, span = 0ull;
    if (mask != 0ull) {
        const int firstp = __builtin_amdgcn_readfirstlane(__builtin_ctzll(mask)), lastp = __builtin_amdgcn_readfirstlane(63 - __builtin_clzll(mask));
        r_lo = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_rel >> 32), firstp) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)my_rel, firstp);
        const unsigned long long r_hi = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(my_rel >> 32), lastp) << 32) |
                                        (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)my_rel, lastp);
        const int s_hi = __builtin_amdgcn_readlane(sl, lastp);
        const unsigned long long stride_hi = ((unsigned long long)s_hi + (kRowAlign - 1)) / kRowAlign * kRowAlign;
        span = r_hi + 4ull * c.dof * stride_hi - r_lo;
    }
    if (lane < np && sl > 0) B.rel[lane] = (unsigned)((my_rel - r_lo) / kRowAlign);
    if (lane == 0) {
        B.rel0 = r_lo;
        B.span = span;
        B.nplans = np;
        B.j0 = j0;
        B.nj = nj;
        B.done = 0;
        B.wide = WIDE ? 1 : 0;
    }
    return true;
}

template <bool STREAMING, typename T, int SEM, bool NV>
LTP_DEV void sample_walk_body(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
                              const unsigned long long* __restrict__ offsets, T* __restrict__ out, unsigned long long capacity, int spread, RowSpec rows,
                              unsigned long long* __restrict__ next_item)
{
    constexpr bool STOP = NV || SEM == kSemMatlab;             // the walk ends at the cap: no end-limit verdict from this launch
    __shared__ WalkBatch buf[kWalkBuffers];
    __shared__ int s_ready[kWalkBuffers];
    __shared__ int s_consumed[kWalkStreamWaves];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x < kWalkBuffers) s_ready[threadIdx.x] = 0;
    if (threadIdx.x < kWalkStreamWaves) s_consumed[threadIdx.x] = 0;
    __syncthreads();
    if (wave < kWalkStreamWaves) {
        // ---- streaming waves: LDS reads and row stores only ----
        for (int seq = 0;; ++seq) {
            const int b = seq % kWalkBuffers;
            while (__hip_atomic_load(&s_ready[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq + 1) __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_readfirstlane(buf[b].done)) break;
            if (walk_long_rows(rows)) walk_stream_rows<STREAMING, T>(buf[b], dof, out, rows, t_sample, wave);
            else if (__builtin_amdgcn_readfirstlane(buf[b].wide)) walk_stream<STREAMING, T, WideSlot>(buf[b], buf[b].wslot, dof, out, rows, t_sample, wave);
            else walk_stream<STREAMING, T, WalkSlot>(buf[b], buf[b].slot, dof, out, rows, t_sample, wave);
            // (release orders the wave's LDS reads of buf[b]; its row stores carry their data in registers)
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&s_consumed[wave], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // ---- builder wave: ordinary loads and LDS stores. A queue item is walk_plans_per_item consecutive plans; its live plans are
    // built walk_plans_per_batch at a time as COMPACT batches; if a lane of one has more than kWalkRuns runs inside the cap (plans that
    // restart mid-motion close to their goal, trajectories that end inside the cap), the same plans are built again as WIDE batches of
    // wpb plans each, every run kept. The walk is one long dependent chain on a SIMD it shares with streaming waves that wait for the
    // memory system anyway: it runs at raised issue priority. ----
    __builtin_amdgcn_s_setprio(3);
    const int lane = (int)(threadIdx.x & 63);
    const WalkQueue queue = walk_queue(count, dof, rows, spread);
    const int ppb = queue.ppb;                                                                    // live plans per compact batch
    const int wpb = kWideLanes / dof > 1 ? kWideLanes / dof : 1;                                  // plans per wide batch
    const int wide_nj = dof < kWideLanes ? dof : kWideLanes;                                      // joints per plan of a wide batch
    const unsigned long long total = queue.total;
    const unsigned long long off0 = offsets[first];
    const int sstride = rows.stride > 1 ? rows.stride : 1;
    const long long needed_end = rows.max_samples > 0 ? (long long)rows.max_samples * sstride : 0x7fffffffffffffffll;   // runs that start at or after this sample are not needed
    const bool long_rows = walk_long_rows(rows);
    const bool no_compact = long_rows || needed_end >= kWalkCompactEnd;                          // (a compact slot's starts have 28 bits)
    int seq = 0;
    auto wait_buffer_free = [&]() {
        if (seq < kWalkBuffers) return;
        const int need = seq - kWalkBuffers + 1;                                                  // every streaming wave past the batch that used this buffer
        for (;;) {
            const int c = __hip_atomic_load(&s_consumed[lane < kWalkStreamWaves ? lane : 0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!__builtin_amdgcn_ballot_w64(c < need)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };
    // the queue is drawn ONE ITEM AHEAD: the atomic's round trip (1-2 us behind the block's own row stores) runs beside the walk
    // of the current item, and so does the load of the next item's trajectory lengths (what its plan list is made from)
    auto draw_issue = [&]() -> unsigned long long {
        unsigned long long item = 0ull;
        if (lane == 0) item = atomicAdd(next_item, 1ull);
        return item;
    };
    auto item_plans = [&](unsigned long long item, long long& pb, int& np) { walk_queue_item(queue, item, pb, np); };
    const WalkCtx ctx{first, count, dof, t_sample, lim, in, rec, offsets, off0, capacity, rows, needed_end, lane};
    // builds and publishes one batch into the buffer the builder has waited for; false (nothing published) if a compact batch does not do
    auto build = [&](long long pb, int plist, int base, int np, int j0, int nj, auto wide_tag) -> bool {
        constexpr bool WIDE = decltype(wide_tag)::value;
        wait_buffer_free();
        if (!walk_build<SEM, WIDE, STOP, kWalkBuilderLean>(ctx, buf[seq % kWalkBuffers], pb, plist, base, np, j0, nj)) return false;
        // publish: everything above is LDS traffic of this one wave, in order
        __hip_atomic_store(&s_ready[seq % kWalkBuffers], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        ++seq;
        return true;
    };
    typedef std::integral_constant<bool, false> CompactTag;
    typedef std::integral_constant<bool, true> WideTag;

    unsigned long long item = walk_uniform(draw_issue());
    long long pb = 0;
    int np_item = 0;
    item_plans(item, pb, np_item);
    int len = walk_item_len_load(ctx, pb, np_item);
    unsigned long long drawn = draw_issue();
    for (;;) {
        if (item >= total) {
            wait_buffer_free();
            if (lane == 0) buf[seq % kWalkBuffers].done = 1;
            __hip_atomic_store(&s_ready[seq % kWalkBuffers], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
        }
        int nlive = 0;
        const int plist = walk_item_plans(ctx, len, np_item, nlive);
        const long long pb_now = pb;
        // the next item: its number has been on its way since the current one started; its lengths travel beside this item's walks
        item = walk_uniform(drawn);
        item_plans(item, pb, np_item);
        len = walk_item_len_load(ctx, pb, np_item);
        drawn = draw_issue();
        for (int base = 0; base < nlive; base += ppb) {
            const int np = nlive - base < ppb ? nlive - base : ppb;
            // (beyond kWalkLanes joints a batch is one plan, taken kWalkLanes joints at a time)
            for (int jc = 0; jc < dof; jc += kWalkLanes) {
                const int jc_end = dof - jc < kWalkLanes ? dof : jc + kWalkLanes;
                if (!no_compact && build(pb_now, plist, base, np, jc, jc_end - jc, CompactTag{})) continue;
                for (int sub = 0; sub < np; sub += wpb)
                    for (int j0 = jc; j0 < jc_end; j0 += wide_nj) {
                        const int npw = np - sub < wpb ? np - sub : wpb, njw = jc_end - j0 < wide_nj ? jc_end - j0 : wide_nj;
                        (void)build(pb_now, plist, base + sub, npw, j0, njw, WideTag{});
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// AUTONOMOUS waves (round 5), for caps of at most kWalkAutoCap samples. The counters (profiles/EXPERIMENTS.md E7.1) showed rows this
// short bound by the builder waves: the kernel above takes a fixed ~2 ms per 1 M plans at every cap from 4 to 16 samples (record
// loads, the walk to the last sample for the end-limit verdict, hand-over) while its streaming waves have next to nothing to write.
// Here every wave is builder AND writer of its own batches: it draws an item, walks its 63 (plan, joint) lanes into its OWN batch
// buffer (walk_build, the same function), and streams the batch's rows itself (walk_stream with one streaming wave); no flags, no
// hand-over, nothing shared between waves but the work queue. Eight waves per block (8 x 19.3 KB of LDS: one block per compute
// unit, two walks per SIMD — six waves of 24.3 KB until the compact slot shrank, E7.9). Rows bit-identical (same functions); plans
// with more runs inside the cap than a compact slot holds are rebuilt as wide batches by the same wave.
// Measured against the builder / streaming-wave form — flags bit 7 — on one box, two rounds (profiles/r05_auto_waves_ab.jsonl: six
// waves against three builders; profiles/r05_four_blocks_ab.txt: eight waves against four builders, sampler kernel in TB/s):
// first-4 0.58 vs 0.55 (both forms of E7.9), first-16 2.42 vs 1.84, first-24 2.63 vs 2.49, first-32 3.46 vs 3.25, receding horizon
// through 32-sample rows 2.68 vs 2.50; first-48 3.40 vs 4.18, first-64 4.36 vs 5.10 — a wave that also writes 10-14 KB of rows per
// plan no longer hides its stores behind another wave's walk — hence the cap of 32. What remains is the walk itself: ~4 500 vector
// instructions per batch (15 divisions of set-up, ~150 vector instructions per run, every run to the last sample for the end-limit
// verdict), i.e. the vector issue rate of the chip (E7.1, E7.6).
// ---------------------------------------------------------------------------------------
constexpr int kWalkAutoWaves = 8;
constexpr int kWalkAutoThreads = kWalkAutoWaves * 64;
#ifndef LTP_WALK_AUTO_CAP
#define LTP_WALK_AUTO_CAP 32
#endif
constexpr int kWalkAutoCap = LTP_WALK_AUTO_CAP;
__host__ __device__ inline bool walk_auto_rows(RowSpec rows)
{
    return rows.max_samples > 0 && rows.max_samples <= kWalkAutoCap && (long long)rows.max_samples * (rows.stride > 1 ? rows.stride : 1) < kWalkCompactEnd;
}

template <bool STREAMING, typename T, int SEM, bool NV>
LTP_DEV void sample_walk_auto_body(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
                                   const unsigned long long* __restrict__ offsets, T* __restrict__ out, unsigned long long capacity, int spread, RowSpec rows,
                                   unsigned long long* __restrict__ next_item)
{
    constexpr bool STOP = NV || SEM == kSemMatlab;
    extern __shared__ __attribute__((aligned(16))) unsigned char ltp_walk_auto_lds[];           // kWalkAutoWaves batch buffers (dynamic: beyond 64 KB)
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = (int)(threadIdx.x & 63);
    WalkBatch& B = reinterpret_cast<WalkBatch*>(ltp_walk_auto_lds)[wave];
    const WalkQueue queue = walk_queue(count, dof, rows, spread);
    const int ppb = queue.ppb;
    const int wpb = kWideLanes / dof > 1 ? kWideLanes / dof : 1;
    const int wide_nj = dof < kWideLanes ? dof : kWideLanes;
    const unsigned long long total = queue.total;
    const int sstride = rows.stride > 1 ? rows.stride : 1;
    const WalkCtx ctx{first, count, dof, t_sample, lim, in, rec, offsets, offsets[first], capacity, rows, (long long)rows.max_samples * sstride, lane};
    auto draw_issue = [&]() -> unsigned long long {
        unsigned long long item = 0ull;
        if (lane == 0) item = atomicAdd(next_item, 1ull);
        return item;
    };
    auto item_plans = [&](unsigned long long item, long long& pb, int& np) { walk_queue_item(queue, item, pb, np); };
    // the wave's batch is complete in LDS (its own stores, in order): write its rows
    auto stream = [&](bool wide) {
        wave_sync();
        if (wide) walk_stream<STREAMING, T, WideSlot>(B, B.wslot, dof, out, rows, t_sample, 0, 1);
        else walk_stream<STREAMING, T, WalkSlot>(B, B.slot, dof, out, rows, t_sample, 0, 1);
        wave_sync();                                                                              // the rows' LDS reads before the next batch's stores
    };
    // one item ahead, as the builder wave above: the next item's number and trajectory lengths travel beside this item's walks
    unsigned long long item = walk_uniform(draw_issue());
    long long pb = 0;
    int np_item = 0;
    item_plans(item, pb, np_item);
    int len = walk_item_len_load(ctx, pb, np_item);
    unsigned long long drawn = draw_issue();
    while (item < total) {
        int nlive = 0;
        const int plist = walk_item_plans(ctx, len, np_item, nlive);
        const long long pb_now = pb;
        item = walk_uniform(drawn);
        item_plans(item, pb, np_item);
        len = walk_item_len_load(ctx, pb, np_item);
        drawn = draw_issue();
        for (int base = 0; base < nlive; base += ppb) {
            const int np = nlive - base < ppb ? nlive - base : ppb;
            for (int jc = 0; jc < dof; jc += kWalkLanes) {
                const int jc_end = dof - jc < kWalkLanes ? dof : jc + kWalkLanes;
                if (walk_build<SEM, false, STOP, kWalkAutoLean>(ctx, B, pb_now, plist, base, np, jc, jc_end - jc)) { stream(false); continue; }
                for (int sub = 0; sub < np; sub += wpb)
                    for (int j0 = jc; j0 < jc_end; j0 += wide_nj) {
                        const int npw = np - sub < wpb ? np - sub : wpb, njw = jc_end - j0 < wide_nj ? jc_end - j0 : wide_nj;
                        (void)walk_build<SEM, true, STOP, kWalkAutoLean>(ctx, B, pb_now, plist, base + sub, npw, j0, njw);
                        stream(true);
                    }
            }
        }
    }
}

#define LTP_WALK_AUTO_KERNEL(NAME, ST, TY, SEM, NV)                                                                                     \
    __global__ void __launch_bounds__(kWalkAutoThreads)                                                                               \
    NAME(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,                             \
         const unsigned long long* __restrict__ offsets, TY* __restrict__ out, unsigned long long capacity, int spread, RowSpec rows, \
         unsigned long long* __restrict__ next_item)                                                                                  \
    {                                                                                                                                 \
        sample_walk_auto_body<ST, TY, SEM, NV>(first, count, dof, t_sample, lim, in, rec, offsets, out, capacity, spread, rows, next_item); \
    }
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f64, false, double, kSemCpp, false)
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f64_nv, false, double, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f64_nt, true, double, kSemCpp, false)
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f64_nt_nv, true, double, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f32, false, float, kSemCpp, false)
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f32_nv, false, float, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f32_nt, true, float, kSemCpp, false)
LTP_WALK_AUTO_KERNEL(k_sample_walk_auto_f32_nt_nv, true, float, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_AUTO_KERNEL(k_sample_walk_matlab_auto_f64, false, double, kSemMatlab, false)
LTP_WALK_AUTO_KERNEL(k_sample_walk_matlab_auto_f64_nt, true, double, kSemMatlab, false)
LTP_WALK_AUTO_KERNEL(k_sample_walk_matlab_auto_f32, false, float, kSemMatlab, false)
LTP_WALK_AUTO_KERNEL(k_sample_walk_matlab_auto_f32_nt, true, float, kSemMatlab, false)
#undef LTP_WALK_AUTO_KERNEL

#define LTP_WALK_KERNEL(NAME, ST, TY, SEM, NV)                                                                                          \
    __global__ void __launch_bounds__(kWalkThreads) __attribute__((amdgpu_waves_per_eu(6, 8)))                                    \
    NAME(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,                             \
         const unsigned long long* __restrict__ offsets, TY* __restrict__ out, unsigned long long capacity, int spread, RowSpec rows, \
         unsigned long long* __restrict__ next_item)                                                                                  \
    {                                                                                                                                 \
        sample_walk_body<ST, TY, SEM, NV>(first, count, dof, t_sample, lim, in, rec, offsets, out, capacity, spread, rows, next_item);     \
    }
LTP_WALK_KERNEL(k_sample_walk_f64, false, double, kSemCpp, false)
LTP_WALK_KERNEL(k_sample_walk_f64_nv, false, double, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_KERNEL(k_sample_walk_f64_nt, true, double, kSemCpp, false)
LTP_WALK_KERNEL(k_sample_walk_f64_nt_nv, true, double, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_KERNEL(k_sample_walk_f32, false, float, kSemCpp, false)
LTP_WALK_KERNEL(k_sample_walk_f32_nv, false, float, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_KERNEL(k_sample_walk_f32_nt, true, float, kSemCpp, false)
LTP_WALK_KERNEL(k_sample_walk_f32_nt_nv, true, float, kSemCpp, true)      // flags bit 4: no end-limit verdict, the walk stops at the cap
LTP_WALK_KERNEL(k_sample_walk_matlab_f64, false, double, kSemMatlab, false)      // LTPlanner.m's sampler (ltp_runs.hpp): same batches, same streaming
LTP_WALK_KERNEL(k_sample_walk_matlab_f64_nt, true, double, kSemMatlab, false)
LTP_WALK_KERNEL(k_sample_walk_matlab_f32, false, float, kSemMatlab, false)
LTP_WALK_KERNEL(k_sample_walk_matlab_f32_nt, true, float, kSemMatlab, false)
#undef LTP_WALK_KERNEL

int sample_walk_resident_blocks(int device, bool f32)
{
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    hipError_t e = f32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample_walk_f32_nt, kWalkThreads, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample_walk_f64_nt, kWalkThreads, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 4;
    return cus * per_cu;
}

// Once per device and handle (reserve(); never inside a stream capture): the autonomous-wave kernels keep kWalkAutoWaves WalkBatch
// records (154 KB) in dynamic LDS, above the default limit — raise it for every variant, checked — and launch one block per
// compute unit. Returns the compute-unit count, or 0 with *err set.
int sample_walk_auto_prepare(int device, hipError_t* err)
{
    const int lds = (int)(kWalkAutoWaves * sizeof(WalkBatch));
    hipError_t e = hipSuccess;
#define LTP_WALK_AUTO_ATTR(K) if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(K), hipFuncAttributeMaxDynamicSharedMemorySize, lds)
    LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f64_nv); LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f64_nt_nv);
    LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f32_nv); LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f32_nt_nv);
    LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f64); LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f64_nt);
    LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f32); LTP_WALK_AUTO_ATTR(k_sample_walk_auto_f32_nt);
    LTP_WALK_AUTO_ATTR(k_sample_walk_matlab_auto_f64); LTP_WALK_AUTO_ATTR(k_sample_walk_matlab_auto_f64_nt);
    LTP_WALK_AUTO_ATTR(k_sample_walk_matlab_auto_f32); LTP_WALK_AUTO_ATTR(k_sample_walk_matlab_auto_f32_nt);
#undef LTP_WALK_AUTO_ATTR
    int cus = 0;
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    if (err) *err = e;
    return e == hipSuccess && cus > 0 ? cus : 0;
}

bool launch_sample_walk(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
                        const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity, int flags, RowSpec rows,
                        unsigned long long* next_item, int resident_blocks, int semantics, int auto_cus)
{
    if (count <= 0) return false;
    // flags bit 4: capped rows without the end-limit verdict — the walk stops at the cap (uncapped rows reach the last sample anyway)
    const int no_verdict = ((flags & 16) && rows.max_samples > 0) ? 1 : 0;
    int spread = (flags >> 8) & 0xFFFF;
    if (spread == 0) spread = kSampleSpread;
    const long long nbatches = walk_queue(count, dof, rows, 1).items;                 // queue items
    if ((long long)spread > nbatches) spread = (int)nbatches;
    if (walk_auto_rows(rows) && !(flags & 128)) {
        // autonomous waves (one block per compute unit); flags bit 7 keeps the builder / streaming-wave form (A/B runs)
        // (the compute-unit count and the kernels' dynamic-LDS limit are per DEVICE and set up once per handle, outside any capture:
        // sample_walk_auto_prepare, called from reserve())
        const unsigned lds = (unsigned)(kWalkAutoWaves * sizeof(WalkBatch));
        const int cus = auto_cus > 0 ? auto_cus : 256;
        long long ablocks = cus;
        if (ablocks * kWalkAutoWaves > nbatches) ablocks = (nbatches + kWalkAutoWaves - 1) / kWalkAutoWaves;
        const dim3 agrid((unsigned)ablocks), ablock(kWalkAutoThreads);
#define LTP_WALK_AUTO_CASE(K, TY)                                                                                                    \
    do {                                                                                                                              \
        hipLaunchKernelGGL(K, agrid, ablock, lds, s, first, count, dof, t_sample, lim, in, rec, offsets, (TY*)out, capacity, spread, rows, next_item); \
    } while (0)
        switch ((flags & 1) | (f32 ? 2 : 0) | (semantics == kSemMatlab ? 4 : (no_verdict ? 8 : 0))) {
        case 8: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f64_nv, double); break;
        case 9: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f64_nt_nv, double); break;
        case 10: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f32_nv, float); break;
        case 11: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f32_nt_nv, float); break;
        case 0: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f64, double); break;
        case 1: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f64_nt, double); break;
        case 2: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f32, float); break;
        case 3: LTP_WALK_AUTO_CASE(k_sample_walk_auto_f32_nt, float); break;
        case 4: LTP_WALK_AUTO_CASE(k_sample_walk_matlab_auto_f64, double); break;
        case 5: LTP_WALK_AUTO_CASE(k_sample_walk_matlab_auto_f64_nt, double); break;
        case 6: LTP_WALK_AUTO_CASE(k_sample_walk_matlab_auto_f32, float); break;
        default: LTP_WALK_AUTO_CASE(k_sample_walk_matlab_auto_f32_nt, float); break;
        }
#undef LTP_WALK_AUTO_CASE
        return true;
    }
    long long blocks = resident_blocks > 0 ? resident_blocks : 1024;
    if (blocks > nbatches) blocks = nbatches;
    const dim3 grid((unsigned)blocks), block(kWalkThreads);
#define LTP_WALK_CASE(K, TY) hipLaunchKernelGGL(K, grid, block, 0, s, first, count, dof, t_sample, lim, in, rec, offsets, (TY*)out, capacity, spread, rows, next_item)
    switch ((flags & 1) | (f32 ? 2 : 0) | (semantics == kSemMatlab ? 4 : (no_verdict ? 8 : 0))) {
    case 8: LTP_WALK_CASE(k_sample_walk_f64_nv, double); break;
    case 9: LTP_WALK_CASE(k_sample_walk_f64_nt_nv, double); break;
    case 10: LTP_WALK_CASE(k_sample_walk_f32_nv, float); break;
    case 11: LTP_WALK_CASE(k_sample_walk_f32_nt_nv, float); break;
    case 0: LTP_WALK_CASE(k_sample_walk_f64, double); break;
    case 1: LTP_WALK_CASE(k_sample_walk_f64_nt, double); break;
    case 2: LTP_WALK_CASE(k_sample_walk_f32, float); break;
    case 3: LTP_WALK_CASE(k_sample_walk_f32_nt, float); break;
    case 4: LTP_WALK_CASE(k_sample_walk_matlab_f64, double); break;
    case 5: LTP_WALK_CASE(k_sample_walk_matlab_f64_nt, double); break;
    case 6: LTP_WALK_CASE(k_sample_walk_matlab_f32, float); break;
    default: LTP_WALK_CASE(k_sample_walk_matlab_f32_nt, float); break;
    }
#undef LTP_WALK_CASE
    return false;
}

}  // namespace ltp
