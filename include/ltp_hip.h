/*
 * ltp_hip.h — C ABI of libltp_hip.so, the MI355X (gfx950) batched planner.
 *
 * This is the drop-in boundary for the reference's hot path
 * LongTermPlanner::planTrajectory and the member functions under it. The
 * reference has no FFI layer of its own (it is one C++ class:
 * /root/reference/include/long_term_planner/long_term_planner.h:61-308), so each
 * entry point below names the reference member it replaces; the drop-in C++
 * class in include/long_term_planner/long_term_planner.h forwards to exactly
 * these symbols and INTEGRATION.md shows the binding.
 *
 * Conventions: plain C, opaque handle, caller-owned buffers, `int` return
 * (0 = LTP_OK, otherwise an ltp_error and ltp_last_error() has the text), no
 * exceptions cross the boundary. `stream` is a hipStream_t (NULL = the default
 * stream). Functions ending in _batch take DEVICE pointers and only enqueue
 * work; functions ending in _host take host pointers and are synchronous.
 * All arithmetic is IEEE binary64, as in the reference.
 *
 * Streams. A handle owns ONE device workspace (compaction queues, lane flags, scan scratch) that
 * ltp_plan_switch_times_batch uses. Calls on one stream are ordered by the stream. When a call arrives on a
 * different stream than the handle's previous workspace user, the library inserts the dependency itself (an event
 * recorded after the previous call, waited for by the new stream), so two streams of one handle — e.g. a
 * double-buffered pipeline — serialise on the workspace instead of racing. For concurrency use one handle per
 * stream (handles are cheap: ~1 MB plus the workspace). A stream that is being captured into a hipGraph is exempt
 * (events cannot cross a capture): do not replay such a graph concurrently with other work of the same handle.
 *
 * Batch geometry. dof, t_sample, max_samples and sample_stride are captured when a batch is planned
 * (ltp_plan_switch_times_batch) and define its records, offsets and row strides. The calls that consume a planned
 * batch (ltp_sample_batch*, ltp_envelope_batch, ltp_build_tables_batch, ltp_replan_states*_batch, ltp_state_at_batch,
 * ltp_end_limit_batch) return LTP_ERR_INVALID_ARGUMENT if one of them was changed on the handle in between.
 */
#ifndef LTP_HIP_H
#define LTP_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ltp_planner ltp_planner;

typedef enum {
    LTP_OK = 0,
    LTP_ERR_INVALID_ARGUMENT = 1,
    LTP_ERR_NO_DEVICE = 2,      /* no HIP device / HIP runtime error: the product has no CPU fallback */
    LTP_ERR_OUT_OF_MEMORY = 3,
    LTP_ERR_HIP = 4
} ltp_error;

/* per-query status bits; LongTermPlanner::planTrajectory's bool is (status == 0) */
#define LTP_STATUS_INVALID_INPUT 1  /* checkInputs false            (src/long_term_planner.cc:14-15, 68-77) */
#define LTP_STATUS_OPT_FAILED    2  /* optSwitchTimes false         (cc:29)                               */
#define LTP_STATUS_NO_SLOWEST    4  /* slowest_joint == -1          (cc:39)                               */
#define LTP_STATUS_END_LIMIT     8  /* last q outside [q_min,q_max] (cc:59-61); trajectory IS filled      */
#define LTP_STATUS_NONFINITE    16  /* defined here: non-finite switching times or a length beyond int   */
                                    /* range; traj_len = 0, not sampled (reference: undefined behaviour)  */
#define LTP_STATUS_OVERFLOW     32  /* trajectory did not fit the output tile passed to ltp_sample_batch  */
#define LTP_STATUS_GOAL_OUTSIDE 64  /* only with ltp_set_goal_check(p, 1): q_goal outside [q_min,q_max]   */
#define LTP_STATUS_MATLAB_ERROR 128 /* only with LTP_SEMANTICS_MATLAB: LTPlanner.m would have raised an error */
                                    /* (checkInputs, index past / vector of filtered roots); traj_len = 0     */
#define LTP_STATUS_MATLAB_COMPLEX 256 /* only with LTP_SEMANTICS_MATLAB, informational: LTPlanner.m would have */
                                    /* carried a complex intermediate; real parts used, the plan IS delivered */

/* Which source the arithmetic follows where the two references diverge:
 * LTP_SEMANTICS_CPP     src/long_term_planner.cc — the parity reference, the default.
 * LTP_SEMANTICS_MATLAB  the MATLAB original LTPlanner.m: polynomial roots picked by POSITION in the output of roots()
 *                       (:346-416) or first-through-a-filter (:247-250, 272-275); optSwitchTimes returns zeros where the C++
 *                       returns false (:222-227, 288-303); fallback rule ~any(t_scaled) (:82); no position limits (no q_0
 *                       check, no end-limit verdict); the slowest joint's jerk-profile flag stays false (:64); sampler with
 *                       1-based correction indices, mod(), cumsum-then-overwrite integration and the acceleration tail zeroed
 *                       for the LAST joint only (:531-624). Rows still carry j as the fourth array. The order of MATLAB's
 *                       roots() is restated from LAPACK's DGEEV path and checked against numpy.roots; against MATLAB itself
 *                       this mode is pinned only by the MATLAB unit tables and grid tests the reference holds. */
#define LTP_SEMANTICS_CPP 0
#define LTP_SEMANTICS_MATLAB 1

/* How the reference's pow(x, 3 | 4 | 6) and pow(x, 1.0 / 2) calls (src/long_term_planner.cc:125-331, 378-621) are formed
 * (pow(x, 2) is x * x in every build: gcc folds it). Both rules are IEEE binary64; they differ in the last bit of about one
 * power in a thousand, which timeScaling's cancelling v_drive formulas (cc:378-446) turn into up to ~5e-11 s of a switching
 * time and the sampler's jerk corrections (cc:768-807) into |dt| * j_max / Ts of single jerk samples.
 * LTP_POW_LIBM   THE DEFAULT. glibc's pow (>= 2.28; the build glibc selects on x86-64 hosts with FMA) restated operation for
 *                operation (csrc/ltp_libm_pow.hpp, bit-identical to the installed libm on 1.7e10 inputs): every record has the
 *                bits a reference built with gcc + glibc computes on such a host and every sample is within 1e-9 of it, no plan
 *                excepted (tests/test_gpu_parity.py; 34.1 M dense trajectories in profiles/r05_parity_report.json). Costs ~45 fp64
 *                operations and 3 table reads per power: +1 % on a full-sampling batch, +10 % on switching times only (round 6: the
 *                powers of the limits are formed once per ltp_set_limits; +35 % before).
 * LTP_POW_EXACT  one rounding of the exact product, sqrt for the power 1/2: what a correctly rounded pow returns; within
 *                1 ulp of ANY libm (so within the differences between two libm builds of the reference); the faster rule. */
#define LTP_POW_EXACT 0
#define LTP_POW_LIBM 1

/* Queries: element (query p, joint j) of each array is ptr[p*query_stride + j*joint_stride].
 * Row-major [n][dof] (the reference's vector-per-query view): query_stride = dof, joint_stride = 1.
 * Joint-major SoA [dof][n]: query_stride = 1, joint_stride = n. */
typedef struct {
    const double* q_goal;
    const double* q_0;
    const double* v_0;
    const double* a_0;
    long long query_stride;
    long long joint_stride;
} ltp_queries;

/* Switching-time records, query-major (what planTrajectory holds in locals, cc:18-24, 31-32, 42). */
typedef struct {
    double* t_opt;        /* [n][dof][7] optimal switch times  (cc:18, 27-30)            */
    double* t_scaled;     /* [n][dof][7] synchronised times after the fallback (cc:20, 43-55) */
    double* dir;          /* [n][dof]    direction of motion   (cc:22)                    */
    double* v_drive;      /* [n][dof]    cruise velocity       (cc:42)                    */
    signed char* mod;     /* [n][dof]    modified-jerk-profile flag (cc:24)               */
    double* t_required;   /* [n]         slowest joint's end time (cc:31)                 */
    int* slowest;         /* [n]         slowest joint, -1 if none (cc:32)                */
    int* traj_len;        /* [n]         Trajectory::length (cc:716-719), 0 if not sampled */
    int* status;          /* [n]         LTP_STATUS_* bits                                 */
} ltp_records;

/* ---- lifetime / configuration -------------------------------------------------------------- */

/* LongTermPlanner::LongTermPlanner(dof, t_sample, q_min, q_max, v_max, a_max, j_max)
 * (long_term_planner.h:118-131). The five arrays hold `dof` doubles. device = HIP ordinal. */
int ltp_create(int dof, double t_sample, const double* q_min, const double* q_max, const double* v_max,
               const double* a_max, const double* j_max, int device, ltp_planner** out);
void ltp_destroy(ltp_planner* p);
/* LongTermPlanner::setLimits (long_term_planner.h:176-187); n_limits = entries per array */
int ltp_set_limits(ltp_planner* p, int n_limits, const double* q_min, const double* q_max, const double* v_max,
                   const double* a_max, const double* j_max);
/* LongTermPlanner::setSampleTime (long_term_planner.h:194-196) */
int ltp_set_sample_time(ltp_planner* p, double t_sample);
/* LongTermPlanner::setDoF (long_term_planner.h:203-205) */
int ltp_set_dof(ltp_planner* p, int dof);
int ltp_get_dof(const ltp_planner* p);
double ltp_get_sample_time(const ltp_planner* p);
const char* ltp_last_error(const ltp_planner* p);
/* name of the kernel that wrote the rows / envelopes of the latest ltp_sample_batch* / ltp_envelope_batch call of this
 * handle ("k_sample", "k_sample_walk_f64_nt", "k_sample_tab_f64_nt", ...: the fused sampler, the walk sampler or the table-pass
 * sampler, see ltp_set_table_pass) */
const char* ltp_last_sampler_kernel(const ltp_planner* p);
/* rows of the packed trajectory layout are padded to this many elements (a multiple of 32) */
int ltp_row_stride(int stored_samples);
/* SURVEY.md §8(f).2 "first N samples only": store min(traj_len, max_samples) samples per row (0 = all of them, which
 * is the reference's behaviour and the default). traj_len in the records stays Trajectory::length; offsets, row
 * strides and the sampler follow the stored length ltp_stored_samples(p, traj_len). The end-limit check (cc:59-61)
 * still refers to the last sample of the whole trajectory. */
int ltp_set_max_samples(ltp_planner* p, int max_samples);
int ltp_get_max_samples(const ltp_planner* p);
int ltp_stored_samples(const ltp_planner* p, int traj_len);
/* SURVEY.md §8(f).2 strided rows: store samples 0, stride, 2*stride, ... (default 1 = every sample); combined with
 * max_samples the cap counts STORED samples. ltp_replan_states_batch's sample index is a stored-sample index. */
int ltp_set_sample_stride(ltp_planner* p, int stride);
int ltp_get_sample_stride(const ltp_planner* p);
/* SURVEY.md §8(f).3 q_goal pre-check, off by default. The reference validates q_0, v_0, a_0 (cc:68-77) but not
 * q_goal: a goal beyond the joint range is planned, sampled and only then reported by the end-limit check (cc:59-61).
 * With enabled != 0 such a query (or a NaN goal) gets LTP_STATUS_GOAL_OUTSIDE, traj_len 0 and is not sampled; every
 * other query is planned exactly as before. */
int ltp_set_goal_check(ltp_planner* p, int enabled);
int ltp_get_goal_check(const ltp_planner* p);
/* LTP_SEMANTICS_CPP (default) or LTP_SEMANTICS_MATLAB, see above. Captured with the batch geometry: the
 * calls that consume a planned batch refuse a handle whose semantics changed in between. With MATLAB semantics rows are
 * written by k_sample_walk_matlab_* (or, on request, the table pass), envelopes take the table pass, single calls the staged
 * path; ltp_end_limit_batch does nothing. */
int ltp_set_semantics(ltp_planner* p, int semantics);
int ltp_get_semantics(const ltp_planner* p);
/* LTP_POW_LIBM (default) or LTP_POW_EXACT, see above. Applies to the calls that plan (switching times, single-joint entry
 * points, the one-launch single call); samplers and consumers form no powers and do not depend on it. */
int ltp_set_pow_rule(ltp_planner* p, int rule);
int ltp_get_pow_rule(const ltp_planner* p);
/* NEW (no counterpart in the reference): which pow rule reproduces the C library THIS process is linked against — the libm a
 * reference built on this host calls at src/long_term_planner.cc:125-331, 378-621. Runs on the host only (no handle, no GPU):
 * `probes` (<= 0: 2^18) planner-sized arguments through the installed pow(x, 3 | 4 | 6 | 1.0 / 2), compared bit for bit with
 * both rules. Returns LTP_POW_LIBM when every result has the restated glibc bits (then the default rule gives the reference's
 * records bit for bit), LTP_POW_EXACT when the installed pow is correctly rounded on all of them (then ltp_set_pow_rule(p,
 * LTP_POW_EXACT) does), or -1: neither — the installed libm is a third one, and about one power in a thousand differs in its
 * last bit from either rule: expect ~2 plans per million with a jerk sample beyond 1e-9 whichever rule is set.
 * The two mismatch counts are written where the pointers are non-null. */
int ltp_host_libm_pow_rule(long long probes, long long* mismatches_libm, long long* mismatches_exact);

/* Where the run tables come from. A sampler / envelope item needs the joint's run tables (<= 20 runs of constant jerk with 10
 * closed-form coefficients each). Three ways, all with bit-identical results:
 *   - built inside the sampler kernel by the item's whole block (k_sample, the envelope's fused form): no extra memory traffic,
 *     ~8 us of latency per item — right for whole float64 rows, which hide it;
 *   - built by one wave of the sampler's block beside its streaming waves (k_sample_walk_*, round 4): no extra memory traffic
 *     either and nothing to hide — right for every row format whose plans have few bytes (capped, float32, sparse rows);
 *   - the table pass: a kernel of their own before the consumer (lane = (plan, joint); 912 bytes per joint through the handle's
 *     workspace) — what ltp_envelope_batch and ltp_build_tables_batch use, and ltp_sample_batch on request (flags bit 2).
 * mode 0 = automatic (rows: k_sample_walk_* for a cap of <= 768 samples, float32 rows, every 3rd sample or sparser, MATLAB
 * semantics, else k_sample; envelopes: the table pass), 1 = never the block-wide fused build (rows: always k_sample_walk_*;
 * envelopes: the table pass), -1 = always the fused build (rows: k_sample; envelopes: built in the kernel). ltp_sample_batch's
 * flags bits 2 / 3 / 5 / 6 choose per call. */
int ltp_set_table_pass(ltp_planner* p, int mode);
int ltp_get_table_pass(const ltp_planner* p);
/* Upper bound (bytes) of the table workspace; default: the larger of 4 GiB and 1/16 of the device's memory (18 GiB on MI355X).
 * Ranges whose tables do not fit are processed in pieces; if the device cannot spare that much, the workspace is smaller. */
int ltp_set_table_workspace(ltp_planner* p, unsigned long long bytes);

/* ---- batched hot path (device pointers, asynchronous on `stream`) -------------------------- */

/* Allocates the handle's device workspace for batches of up to n queries now. The batched calls below grow it on
 * demand (hipMalloc / hipFree), which is not allowed while `stream` is being captured into a hipGraph: call this once
 * before hipStreamBeginCapture — and ltp_reserve_tables if the captured calls take the table pass — then
 * ltp_plan_switch_times_batch / ltp_sample_batch / ltp_envelope_batch / ltp_replan_states_batch only enqueue memset and
 * kernel nodes and the graph can be replayed on new inputs in place. Growing a workspace later (a larger n, a larger
 * ltp_reserve_tables) frees the old buffers: graphs instantiated before that must not be replayed any more. */
int ltp_reserve_batch(ltp_planner* p, long long n);
/* Allocates the table-pass workspace for ranges of up to n plans (at most ltp_set_table_workspace bytes; longer ranges
 * are processed in pieces). Needed before capturing a call that takes the table pass — ltp_envelope_batch by default,
 * ltp_build_tables_batch's callers with the library's workspace, ltp_sample_batch* when flag bit 2 is set, or when flag bit 5 forbids
 * the walk kernel for rows of at most 8 KB (float64) / 16 KB (float32) per joint under a cap (those then take the table pass), and
 * every ltp_sample_batch* call in MATLAB semantics with more than the walk kernel forbidden: while a stream is
 * being captured the library neither allocates nor frees; it cuts the range into pieces that fit the workspace it has
 * and returns LTP_ERR_INVALID_ARGUMENT if it has none. */
int ltp_reserve_tables(ltp_planner* p, long long n);

/* planTrajectory stages 1-3 for n queries (cc:14-55): checkInputs, optSwitchTimes per joint,
 * slowest-joint reduction, timeScaling per other joint, fallback copy; then traj_len (cc:716-719).
 * offsets (device, [n+1], may be NULL) receives the exclusive scan of the packed trajectory sizes
 * in elements (doubles, or floats for ltp_sample_batch_f32): plan p occupies [offsets[p], offsets[p+1]) of a packed
 * buffer, laid out [q,v,a,j][joint][ltp_row_stride(stored samples of p)]. */
int ltp_plan_switch_times_batch(ltp_planner* p, long long n, const ltp_queries* in, const ltp_records* out,
                                unsigned long long* offsets, void* stream);

/* The end-limit check of planTrajectory (cc:59-61) WITHOUT sampling: for plans [first, first+count) of a planned
 * batch, walks every joint's runs to the last trajectory sample (the value ltp_sample_batch would store at
 * traj_len-1, bit for bit) and sets LTP_STATUS_END_LIMIT where it lies outside [q_min, q_max]. After this call
 * status == 0 is exactly planTrajectory's return value. ltp_sample_batch* and ltp_envelope_batch apply the same
 * check themselves; ltp_plan_switch_times_batch alone does not (status then holds the pre-sampling verdict
 * cc:14-39 only). The _host calls that do not sample run it for the caller. */
int ltp_end_limit_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                        void* stream);

/* getTrajectory (cc:706-841) + the end-limit check (cc:59-61) for plans [first, first+count):
 * plan p is written at out + (offsets[p] - offsets[first]); plans that would end beyond
 * `capacity` ELEMENTS get LTP_STATUS_OVERFLOW and are skipped. offsets, row strides and capacity are in elements and identical
 * for both row formats (rows are padded to 32 elements). `out`: device, 16-byte aligned.
 *
 * ltp_sample_batch_ex is the entry point; its policy is a struct of named fields. Zero-initialise it, set .size =
 * sizeof(ltp_sample_opts) and only the fields you mean (opts == NULL: all defaults). Every choice writes THE SAME ROWS; the
 * fields only pick which kernel does it (DESIGN.md §4 has the measurements behind the defaults). */
#define LTP_ROWS_F64 0               /* binary64 rows (the reference's) */
#define LTP_ROWS_F32 1               /* the same binary64 results, rounded once to float when stored */
#define LTP_STORES_NONTEMPORAL 0     /* default: rows bypass the caches (they are written once and read by someone else) */
#define LTP_STORES_PLAIN 1
#define LTP_SAMPLER_AUTO 0           /* default: the library picks by row format (below) */
#define LTP_SAMPLER_FUSED 1          /* k_sample_*: a block builds a plan's run tables in LDS and streams its rows (whole f64 rows take it) */
#define LTP_SAMPLER_WALK 2           /* k_sample_walk_*: builder wave + streaming waves, tables never leave the compute unit (capped rows of
                                      * <= 768 samples, float32 rows, rows of every 3rd sample or sparser, MATLAB semantics take it);
                                      * caps of <= 32 samples use its autonomous-wave form (every wave builds and writes its own batches) */
#define LTP_SAMPLER_WALK_STREAMING 3 /* ... and keep the builder / streaming-wave form for caps of <= 32 samples too (A/B runs) */
#define LTP_SAMPLER_TABLE 4          /* k_build_tables + k_sample_tab_*: through the packed run tables of include/ltp_run_tables.hpp in HBM */
#define LTP_VERDICT_KEEP 0           /* default: the call also forms planTrajectory's end-limit verdict (cc:59-61, LTP_STATUS_END_LIMIT) */
#define LTP_VERDICT_SKIP 1           /* capped rows: stop at the cap instead of walking every joint to its last sample; LTP_STATUS_END_LIMIT
                                      * is then left unspecified by this call (ltp_end_limit_batch / end_limit = 1 form it); same rows */
typedef struct {
    unsigned size;       /* sizeof(ltp_sample_opts) in the caller's build: fields beyond it keep their defaults */
    int format;          /* LTP_ROWS_* : the element type `out` points to */
    int stores;          /* LTP_STORES_* */
    int sampler;         /* LTP_SAMPLER_* */
    int verdict;         /* LTP_VERDICT_* */
    int interleave;      /* block interleave factor of the work queue: 0 = default (64), 1 = blocks in plan order, <= 65535. Large
                          * tiles (>= 64 GiB) written with the default reach the HBM fill ceiling */
    int dry_run;         /* DIAGNOSTIC, 0 | 1: the store pattern without the arithmetic (not results) */
} ltp_sample_opts;
int ltp_sample_batch_ex(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                        const unsigned long long* offsets, void* out, unsigned long long capacity, const ltp_sample_opts* opts,
                        void* stream);
/* The same call with the policy packed into an int (rounds 1-5; kept for its callers, a thin wrapper over the same launcher):
 * bit 0 = LTP_STORES_NONTEMPORAL (NOTE: 0 here means plain stores), bit 1 = dry_run, bit 2 = LTP_SAMPLER_TABLE, bit 3 =
 * LTP_SAMPLER_FUSED, bit 4 = LTP_VERDICT_SKIP, bit 5 = never the walk kernels, bit 6 = LTP_SAMPLER_WALK, bit 6 | bit 7 =
 * LTP_SAMPLER_WALK_STREAMING, bits 8..23 = interleave. New code: ltp_sample_batch_ex. */
int ltp_sample_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                     const unsigned long long* offsets, double* out, unsigned long long capacity, int flags, void* stream);
int ltp_sample_batch_f32(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                         const unsigned long long* offsets, float* out, unsigned long long capacity, int flags, void* stream);

/* SURVEY.md §8(f).2 on-device consumer: position envelopes instead of dense rows. For plans [first, first+count) of
 * a batch planned by ltp_plan_switch_times_batch, env[((i*dof + j)*n_windows + w)*2 + {0,1}] = {min, max} of the q
 * samples w*window .. (w+1)*window-1 of joint j of local plan i — exactly the values getTrajectory (cc:706-841) /
 * ltp_sample_batch would have stored, reduced on the fly, so the dense trajectories (32*dof*traj_len bytes per plan)
 * never exist. Windows that start at or after traj_len hold the last position twice; plans with traj_len 0 (failed or
 * rejected) hold NaN. The end-limit check (cc:59-61) sets LTP_STATUS_END_LIMIT as the sampler does. max_samples and
 * sample_stride do not apply. env: device, count*dof*n_windows*2 doubles, 16-byte aligned. */
int ltp_envelope_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                       int window, int n_windows, double* env, void* stream);
/* How the envelope calls (ltp_envelope_batch, ltp_envelope_multi, ltp_plan_envelope*_host) find a window's extreme samples.
 * LTP_ENVELOPE_ANALYTIC (default)    inside a run q is one cubic in the sample index, so only the samples at the ends of each (run,
 *                                    window) stretch and either side of the real roots of its derivative are evaluated (they ARE samples
 *                                    of the row): a few evaluations per run instead of `window`, by a lane-per-(plan, joint) walk without
 *                                    run tables or workspace (k_envelope_walk: 3x the plans/s). By construction within a few ulps of q
 *                                    (<= 1e-12) of the exhaustive result — it could differ only where a neighbouring sample undercuts by
 *                                    rounding alone; MEASURED identical in all 8.8e9 window values of profiles/r06_envelope_mode_soak.json
 *                                    (panda, the reference's limits, 30-DoF, 36 wide-fuzzed limit sets, six window geometries), which is
 *                                    why it is the default since round 6. With ltp_set_table_pass(p, 1 | -1) the block-cooperative
 *                                    kernel's analytic form runs instead (same values).
 * LTP_ENVELOPE_EXHAUSTIVE            every sample of the window is evaluated: min / max have the BITS of the same reduction of the rows,
 *                                    by construction. The opt-in for callers who need that guarantee rather than the measurement. */
#define LTP_ENVELOPE_EXHAUSTIVE 0
#define LTP_ENVELOPE_ANALYTIC 1
int ltp_set_envelope_mode(ltp_planner* p, int mode);
int ltp_get_envelope_mode(const ltp_planner* p);

/* SURVEY.md §8(f).2 consumer HOOK: the run tables of plans [first, first+count) of a planned batch in a CALLER buffer, for
 * consumers of the caller's own (include/ltp_run_tables.hpp: the format, run_coef / run_eval, for_each_run / for_each_sample, and
 * the block-cooperative LDS form — the same single-source device functions ltp_sample_batch and ltp_envelope_batch are built
 * from). Per (plan, joint) 912 bytes: the <= 20 runs into which getTrajectory's jerk array (cc:735-807) and snap rules
 * (cc:815-829) cut the trajectory, each with the state before it; every sample is a closed-form polynomial of its position in
 * its run, with exactly the bits ltp_sample_batch would have stored. Lane index of (plan, joint) = (plan - first) * dof + joint.
 * tables: device, 16-byte aligned, bytes >= ltp_run_tables_bytes(p, count) (whole tiles of 64 lanes; the lanes of the last tile
 * beyond count * dof are not written). Plans with traj_len 0 get runs == 0. Applies the end-limit check (cc:59-61,
 * LTP_STATUS_END_LIMIT) like the samplers do (C++ semantics). max_samples / sample_stride do not apply: tables always cover the
 * whole trajectory. tests/cpp/example_consumer.hip is a complete user-side consumer. */
unsigned long long ltp_run_tables_bytes(const ltp_planner* p, long long n_plans);
int ltp_build_tables_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                           unsigned long long* tables, unsigned long long bytes, void* stream);

/* SURVEY.md §8(f).1 receding horizon (reference README.md:10-13): start states of the next plans = sample k of the
 * trajectories sampled into `tile` by ltp_sample_batch(first, count, ...). sample_index: device int[count] or NULL
 * (then uniform_index for all); k is clamped to the stored samples; plans that were not sampled — traj_len 0,
 * LTP_STATUS_OVERFLOW, or rows that would end beyond `capacity` elements of `tile` (the capacity given to
 * ltp_sample_batch) — keep the start state they had in `in`; nothing outside the tile is read. Output element (local plan i, joint j) at
 * ptr[i*query_stride + j*joint_stride]. The float64 form IGNORES THE CONTENT of `tile` (only its capacity rule applies; the pointer
 * must still be non-NULL): a stored float64 sample has the bits of the closed-form run evaluation, so the state is recomputed from
 * the records (a quarter of the time of 8-byte gathers from the tile). It therefore returns what ltp_sample_batch WOULD have stored:
 * `rec` and `in` must be unchanged since the batch was planned, and a tile that was written by a dry run (flags bit 1), edited by the
 * caller or never sampled is not noticed. Callers that post-process the tile and want the edited values use the float32 form (which
 * reads the ROUNDED values the tile holds) or gather themselves. */
int ltp_replan_states_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                            const unsigned long long* offsets, const double* tile, unsigned long long capacity,
                            const int* sample_index, int uniform_index,
                            double* q_0, double* v_0, double* a_0, long long query_stride, long long joint_stride, void* stream);
int ltp_replan_states_f32_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                                const unsigned long long* offsets, const float* tile, unsigned long long capacity,
                                const int* sample_index, int uniform_index,
                                double* q_0, double* v_0, double* a_0, long long query_stride, long long joint_stride, void* stream);
/* The same without any sampled rows: the state at TRAJECTORY sample k (0 .. traj_len-1, clamped; not a stored-sample
 * index) of plans [first, first+count), computed from the switching-time records alone with the sampler's own run
 * tables, i.e. with the bits ltp_sample_batch would have stored at k. No tile, no offsets: the cheap way to close a
 * receding-horizon loop on the device when only the restart state is needed. */
int ltp_state_at_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                       const int* sample_index, int uniform_index, double* q_0, double* v_0, double* a_0,
                       long long query_stride, long long joint_stride, void* stream);

/* Synthetic queries of SURVEY.md §8(d) (distribution of tests/randomConfiguration.m:14-34 with per-joint
 * limits), counter-based: query index first_query+p, so shards of one batch can be generated anywhere. */
int ltp_generate_queries_batch(ltp_planner* p, long long n, unsigned long long seed, long long first_query,
                               double* q_goal, double* q_0, double* v_0, double* a_0,
                               long long query_stride, long long joint_stride, void* stream);

/* ---- host-pointer convenience (synchronous) ------------------------------------------------- */

/* Full planTrajectory (cc:7-63) for n row-major [n][dof] host queries. Any record pointer may be NULL.
 * If packed != NULL, *packed receives a malloc'ed buffer of offsets[n] doubles (free with ltp_free_host)
 * and offsets ([n+1], host) must be non-NULL. With packed == NULL nothing is sampled, but status still carries
 * LTP_STATUS_END_LIMIT (ltp_end_limit_batch runs instead of the sampler): status == 0 is planTrajectory's bool. */
int ltp_plan_batch_host(ltp_planner* p, long long n, const double* q_goal, const double* q_0, const double* v_0,
                        const double* a_0, const ltp_records* host_records, unsigned long long* offsets,
                        double** packed);

/* ---- one process, several devices: contiguous query ranges, no collective ---------------- */

/* Range of shard `rank` of `world` over n queries: [*first, *first + *count), remainder to the lowest ranks. The same
 * rule as longtermplanner_amd/parallel.py::shard_range, bench.py and planTrajectoryBatchSharded. */
void ltp_shard_range(long long n, int rank, int world, long long* first, long long* count);

/* ltp_plan_batch_host over k planners, normally one per device (ltp_create(..., device = g, ...)), all configured
 * identically (else LTP_ERR_INVALID_ARGUMENT): shard g plans queries ltp_shard_range(n, g, k) on its own device from
 * its own host thread, all shards concurrently; there is no exchange between devices (queries are independent; the
 * only cross-lane step of planTrajectory, the slowest-joint reduction cc:31-39, is inside a query). Results are
 * written as ONE batch: records at their global query index, offsets rebased to the concatenated `*packed` buffer —
 * bit-identical to a single ltp_plan_batch_host call over all n queries. Several planners may name the same device
 * ("virtual shards", how the tests check this on one GPU). On error the first failing shard's code is returned
 * and ltp_last_error(planners[0]) names the shard. */
int ltp_plan_batch_multi(ltp_planner* const* planners, int k, long long n, const double* q_goal, const double* q_0,
                         const double* v_0, const double* a_0, const ltp_records* host_records, unsigned long long* offsets,
                         double** packed);

/* The same sharding with DEVICE-RESIDENT shards: nothing passes through the host. Shard g = queries ltp_shard_range(n, g, k)
 * of one batch of n; its queries, records and offsets live on planners[g]'s device as SHARD-LOCAL arrays (element 0 is the
 * shard's first query). One host thread per shard enqueues the work on shards[g].stream; the calls return when everything
 * is enqueued (asynchronous like the _batch calls; ltp_synchronize_multi waits for all shards). Results are bit-identical
 * to the single-handle calls over the whole batch, shard by shard. Planners as for ltp_plan_batch_multi (distinct
 * handles, identically configured, devices may repeat). */
typedef struct {
    ltp_queries in;                /* device pointers, shard-local */
    ltp_records out;               /* device pointers, shard-local, all non-NULL */
    unsigned long long* offsets;   /* device [count + 1] or NULL */
    void* stream;                  /* hipStream_t on planners[g]'s device, NULL = its default stream */
} ltp_shard;
/* ltp_plan_switch_times_batch per shard; end_limit != 0 adds ltp_end_limit_batch (status == 0 is then planTrajectory's bool) */
int ltp_plan_switch_times_multi(ltp_planner* const* planners, int k, long long n, const ltp_shard* shards, int end_limit);
/* ltp_envelope_batch per shard of a batch planned by ltp_plan_switch_times_multi; env[g]: device, count_g*dof*n_windows*2 doubles */
int ltp_envelope_multi(ltp_planner* const* planners, int k, long long n, const ltp_shard* shards, int window, int n_windows,
                       double* const* env);
/* ltp_state_at_batch per shard (receding horizon without rows). sample_index: NULL or per-shard device int arrays (entries may
 * be NULL); q_0[g], v_0[g], a_0[g]: device arrays laid out like the shard's queries (same strides) */
int ltp_state_at_multi(ltp_planner* const* planners, int k, long long n, const ltp_shard* shards, const int* const* sample_index,
                       int uniform_index, double* const* q_0, double* const* v_0, double* const* a_0);
int ltp_synchronize_multi(ltp_planner* const* planners, int k, const ltp_shard* shards);
/* ltp_plan_envelope_host over k planners: host arrays in, host envelopes and records out at their global query index
 * (what LongTermPlanner::planEnvelopeBatchSharded calls) */
int ltp_plan_envelope_multi_host(ltp_planner* const* planners, int k, long long n, const double* q_goal, const double* q_0,
                                 const double* v_0, const double* a_0, int window, int n_windows, const ltp_records* host_records,
                                 double* env);

/* planTrajectory stages 1-3 + the on-device envelope consumer (ltp_envelope_batch) for host arrays: a host caller
 * cannot take in the dense trajectories of a large batch (32*dof*traj_len bytes per plan over PCIe), but it can take
 * their position envelopes. env: host, n*dof*n_windows*2 doubles, layout as in ltp_envelope_batch. host_records
 * (optional, members may be NULL) receives the records as ltp_plan_batch_host does; status includes END_LIMIT. */
int ltp_plan_envelope_host(ltp_planner* p, long long n, const double* q_goal, const double* q_0, const double* v_0,
                           const double* a_0, int window, int n_windows, const ltp_records* host_records, double* env);

/* LongTermPlanner::getTrajectory (cc:706-841) for n host records ([n][dof][7] times etc.). */
int ltp_get_trajectory_host(ltp_planner* p, long long n, const double* t, const double* dir, const signed char* mod,
                            const double* q_0, const double* v_0, const double* a_0, const double* v_drive,
                            int* traj_len, int* status, unsigned long long* offsets, double** packed);
void ltp_free_host(void* ptr);

/* LongTermPlanner::checkInputs (cc:68-77) */
int ltp_check_inputs_host(ltp_planner* p, const double* q_0, const double* v_0, const double* a_0, int* ok);
/* LongTermPlanner::optBraking (cc:650-701); t_rel[7] is in/out (only [0..2] are written) */
int ltp_opt_braking_host(ltp_planner* p, int joint, double v_0, double a_0, double* q, double* t_rel, double* dir);
/* LongTermPlanner::optSwitchTimes (cc:82-353); t[7] is in/out (written only where the reference writes it) */
int ltp_opt_switch_times_host(ltp_planner* p, int joint, double q_goal, double q_0, double v_0, double a_0, double v_drive,
                              double* t, double* dir, char* mod, int* ok);
/* LongTermPlanner::timeScaling (cc:358-645); accepted_case (may be NULL): 1..8, 0 = none */
int ltp_time_scaling_host(ltp_planner* p, int joint, double q_goal, double q_0, double v_0, double a_0, double dir,
                          double t_required, double* scaled_t, double* v_drive, char* mod, int* ok, int* accepted_case);

/* roots<T>() of the reference's long_term_planner/roots.h:22-34 (the one third-party computation of the hot path: Eigen 3.4's
 * EigenSolver on the monic companion matrix) for n polynomials of degree 1..8: ALL eigenvalues, in Eigen's output order and
 * conjugate-pair convention ((re, +im) first), real eigenvalues with an exactly zero imaginary part — what
 * getSmallestPositiveNonComplexRoot (roots.h:43-50) selects from. coef: [n][degree+1], highest coefficient first; re, im:
 * [n][degree]. A non-finite companion matrix or a non-converged iteration gives NaN (the reference: uninitialised).
 * The float form exists because the reference's own known-answer test is in float (tests/src/roots_tests.cc:9-32). */
int ltp_roots_f64_host(ltp_planner* p, long long n, int degree, const double* coef, double* re, double* im);
int ltp_roots_f32_host(ltp_planner* p, long long n, int degree, const float* coef, float* re, float* im);

/* ---- diagnostics used by the parity tests ---------------------------------------------------- */
/* MATLAB semantics: flags of the latest one-lane call (ltp_opt_*_host, ltp_time_scaling_host): 1 = complex intermediate, 2 = LTPlanner.m would have raised an error */
int ltp_debug_last_matlab_flags(const ltp_planner* p);
/* MATLAB's roots() as the MATLAB-semantics kernels compute it, n polynomials of degree 1..6: re, im [n][degree] in MATLAB's
 * output order, nroots [n] (degree minus stripped leading zeros), status [n] (0 ok, 1 no convergence, 2 NaN / Inf) */
int ltp_debug_roots_matlab_host(ltp_planner* p, long long n, int degree, const double* coef, double* re, double* im, int* nroots, int* status);
/* device_buffer (3 x count u64, or NULL to switch off): k_sample block start / run tables ready / end on the
 * 100 MHz wall clock; ltp_envelope_batch writes 16 u64 per (plan, joint group) item instead: loop top, item drawn,
 * traj_len read, after each of the seven table-build barriers, reduction done */
int ltp_debug_set_sample_stamps(ltp_planner* p, unsigned long long* device_buffer);
/* tuning aid: size of the persistent k_sample / k_envelope grid (0 = what the device holds at once, the default) */
int ltp_debug_set_sample_blocks(ltp_planner* p, int blocks);
/* resident blocks of the persistent grids: which = 0 k_sample float64, 1 k_sample float32, 2 k_envelope, 3 / 4 k_sample_tab float64 / float32 */
int ltp_debug_get_sample_blocks(ltp_planner* p, int which);
/* out[i*8 + {0..7}] = x/y, sqrt|x|, x^3, x^4, x^6, floor(x/y), ceil(x/y), x*y+x computed on the device */
int ltp_debug_math_probe_host(ltp_planner* p, long long n, const double* x, const double* y, double* out);
/* out[i] = pow(x[i], y[i]) by the restated glibc pow of LTP_POW_LIBM (csrc/ltp_libm_pow.hpp), any finite or non-finite x, y */
int ltp_debug_libm_pow_host(ltp_planner* p, long long n, const double* x, const double* y, double* out);
/* root[i] = smallest positive exactly-real root of the degree-`degree` polynomial coef[i*7 .. i*7+degree]
 * (roots.h:22-50 semantics) */
int ltp_debug_roots_probe_host(ltp_planner* p, long long n, int degree, const double* coef, double* root);

#ifdef __cplusplus
}
#endif
#endif /* LTP_HIP_H */
