// ltp_capi_batch.hip — C ABI (include/ltp_hip.h): the batched hot path on device pointers (asynchronous on the caller's stream).
#include "ltp_handle.hpp"

using namespace ltp_capi;

extern "C" {

int ltp_plan_switch_times_batch(ltp_planner* p, long long n, const ltp_queries* in, const ltp_records* out,
                                unsigned long long* offsets, void* stream)
{
    if (!p || n < 0 || !in || !records_complete(out)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc != LTP_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    if (n == 0 || p->dof == 0) {
        capture_geometry(p);
        // dof == 0: every query fails with slowest_joint == -1 (cc:39); nothing to launch per joint
        if (offsets) LTP_HIP_TRY(p, hipMemsetAsync(offsets, 0, sizeof(unsigned long long) * (size_t)(n + 1), s));
        if (n > 0) {
            std::vector<int> st((size_t)n, LTP_STATUS_NO_SLOWEST), neg((size_t)n, -1);
            std::vector<double> tr((size_t)n, -1.0);
            LTP_HIP_TRY(p, hipMemcpyAsync(out->status, st.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
            LTP_HIP_TRY(p, hipMemcpyAsync(out->slowest, neg.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, s));
            LTP_HIP_TRY(p, hipMemcpyAsync(out->t_required, tr.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, s));
            LTP_HIP_TRY(p, hipMemsetAsync(out->traj_len, 0, sizeof(int) * (size_t)n, s));
            LTP_HIP_TRY(p, hipStreamSynchronize(s));
        }
        return LTP_OK;
    }
    rc = reserve(p, n);
    if (rc != LTP_OK) return rc;
    bool capturing = false;
    if ((rc = workspace_acquire(p, s, capturing)) != LTP_OK) return rc;
    capture_geometry(p);
    const ltp::Limits L = dev_limits(p);
    const ltp::Queries q = to_dev(in);
    const ltp::Records r = to_dev(out);
    LTP_HIP_TRY(p, hipMemsetAsync(p->d_queue_count, 0, 16 * sizeof(unsigned long long), s));
    ltp::launch_switch_times(s, n, p->dof, p->t_sample, p->goal_check, L, q, r, p->d_lane_flags, p->d_queue, p->d_queue_count, stage_variant(p));
    ltp::launch_offsets(s, n, p->dof, p->t_sample, r, p->d_block_sums, offsets ? offsets : p->d_offsets_scratch, true, ltp::RowSpec{p->max_samples, p->sample_stride});
    LTP_HIP_TRY(p, hipGetLastError());
    return workspace_release(p, s, capturing);
}

int ltp_end_limit_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec, void* stream)
{
    if (!p || first < 0 || count < 0 || !in || !records_complete(rec)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc == LTP_OK) rc = check_geometry(p);
    if (rc != LTP_OK) return rc;
    if (p->semantics == LTP_SEMANTICS_MATLAB) return LTP_OK;   // LTPlanner.m has no position limits: there is no end-limit verdict
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    ltp::launch_end_limit((hipStream_t)stream, first, count, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec));
    LTP_HIP_TRY(p, hipGetLastError());
    return LTP_OK;
}

static int sample_batch_any(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                            const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity, int flags,
                            void* stream)
{
    if (!p || first < 0 || count < 0 || !in || !records_complete(rec) || !offsets || (!out && capacity > 0))
        return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    if (((uintptr_t)out & 15u) != 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "trajectory buffer must be 16-byte aligned");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc == LTP_OK) rc = check_geometry(p);
    if (rc != LTP_OK) return rc;
    if (count == 0 || p->dof == 0) return LTP_OK;
    if ((rc = reserve(p, 0)) != LTP_OK) return rc;   // work-queue heads, resident block counts (no-op after the first call)
    const hipStream_t s = (hipStream_t)stream;
    const ltp::RowSpec rows{p->max_samples, p->sample_stride};
    const int blocks = p->sample_blocks_override > 0 ? p->sample_blocks_override : p->sample_blocks[f32 ? 1 : 0];
    // bytes of one joint's four rows when the cap applies (a cap is the only way rows are known to be short up front)
    const unsigned long long row_bytes = p->max_samples > 0 ? 4ull * (f32 ? 4 : 8) * (unsigned long long)p->max_samples : 0ull;
    // MATLAB semantics: the fused build of k_sample exists for the C++ semantics only; the walk kernel and the table pass (whose
    // builders are for_each_run<SEM>) serve both, and the kernels that read tables do not depend on the semantics
    const bool matlab = p->semantics == LTP_SEMANTICS_MATLAB;
    // k_sample_walk_* — the tables stay in the compute unit, no table pass at all. Taken by itself for the rows want_walk() names
    // (capped, float32, sparse) and for every row format in MATLAB semantics; flags bit 6 forces it, bit 5 forbids it, bit 2 = "the
    // table-pass kernels" and bit 3 = "the fused build" keep their meaning. (Diagnostic runs — dry stores, stamps — stay with the
    // kernels that implement them.)
    if (!p->dbg_stamps && !(flags & (2 | 32)) && ltp::sample_walk_applies(p->dof, rows) &&
        ((flags & 64) || (!(flags & (4 | 8)) && (matlab || want_walk(p, rows.max_samples, rows.stride, f32))))) {
        if (p->walk_blocks[f32 ? 1 : 0] == 0) p->walk_blocks[f32 ? 1 : 0] = ltp::sample_walk_resident_blocks(p->device, f32);
        if (p->walk_auto_cus == 0) {       // normally done by reserve() when the batch was planned
            hipError_t e = hipSuccess;
            p->walk_auto_cus = ltp::sample_walk_auto_prepare(p->device, &e);
            LTP_HIP_TRY(p, e);
        }
        unsigned long long* head = p->d_sample_next + (p->sample_next_slot++ & 63u);
        LTP_HIP_TRY(p, hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
        const bool autonomous = ltp::launch_sample_walk(s, first, count, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), offsets, out, f32, capacity, flags, rows, head,
                                p->sample_blocks_override > 0 ? p->sample_blocks_override : p->walk_blocks[f32 ? 1 : 0], matlab ? ltp::kSemMatlab : ltp::kSemCpp, p->walk_auto_cus);
        LTP_HIP_TRY(p, hipGetLastError());
        const bool nv = !matlab && (flags & 16) && rows.max_samples > 0;     // flags bit 4: capped rows without the end-limit verdict
        if (nv)
            p->last_kernel = autonomous ? (f32 ? ((flags & 1) ? "k_sample_walk_auto_f32_nt_nv" : "k_sample_walk_auto_f32_nv") : ((flags & 1) ? "k_sample_walk_auto_f64_nt_nv" : "k_sample_walk_auto_f64_nv"))
                                        : (f32 ? ((flags & 1) ? "k_sample_walk_f32_nt_nv" : "k_sample_walk_f32_nv") : ((flags & 1) ? "k_sample_walk_f64_nt_nv" : "k_sample_walk_f64_nv"));
        else if (autonomous)     // caps of at most 32 samples: every wave builds and writes its own batches (ltp_sampler_walk.hip)
            p->last_kernel = matlab ? (f32 ? ((flags & 1) ? "k_sample_walk_matlab_auto_f32_nt" : "k_sample_walk_matlab_auto_f32") : ((flags & 1) ? "k_sample_walk_matlab_auto_f64_nt" : "k_sample_walk_matlab_auto_f64"))
                                    : (f32 ? ((flags & 1) ? "k_sample_walk_auto_f32_nt" : "k_sample_walk_auto_f32") : ((flags & 1) ? "k_sample_walk_auto_f64_nt" : "k_sample_walk_auto_f64"));
        else
        p->last_kernel = matlab ? (f32 ? ((flags & 1) ? "k_sample_walk_matlab_f32_nt" : "k_sample_walk_matlab_f32") : ((flags & 1) ? "k_sample_walk_matlab_f64_nt" : "k_sample_walk_matlab_f64"))
                                : (f32 ? ((flags & 1) ? "k_sample_walk_f32_nt" : "k_sample_walk_f32") : ((flags & 1) ? "k_sample_walk_f64_nt" : "k_sample_walk_f64"));
        return LTP_OK;
    }
    if (matlab || (!(flags & 2) && (!p->dbg_stamps || (flags & 4)) && ((flags & 4) || (!(flags & 8) && want_table_pass(p, row_bytes, f32))))) {
        // table pass: per piece of the range, k_build_tables then the sampler variant that reads the tables
        bool capturing = false;
        if ((rc = workspace_acquire(p, s, capturing)) != LTP_OK) return rc;
        long long piece = 0;
        if ((rc = ensure_tables(p, count, capturing, &piece)) != LTP_OK) return rc;
        for (long long f = first; f < first + count; f += piece) {
            const long long c = first + count - f < piece ? first + count - f : piece;
            unsigned long long* head = p->d_sample_next + (p->sample_next_slot++ & 63u);
            LTP_HIP_TRY(p, hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
            ltp::launch_build_tables(s, f, c, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), rows, false, offsets, first, p->d_tables, p->semantics);
            ltp::launch_sample_tab(s, f, c, first, p->dof, to_dev(rec), offsets, out, f32, capacity, flags & ~2, rows, head,
                                   p->sample_blocks_override > 0 ? p->sample_blocks_override : p->sample_blocks[f32 ? 4 : 3], p->d_tables, p->t_sample, p->dbg_stamps);
        }
        LTP_HIP_TRY(p, hipGetLastError());
        p->last_kernel = f32 ? ((flags & 1) ? "k_sample_tab_f32_nt" : "k_sample_tab_f32") : ((flags & 1) ? "k_sample_tab_f64_nt" : "k_sample_tab_f64");
        return workspace_release(p, s, capturing);
    }
    // each launch gets its own work-queue head from a ring of 64, zeroed in stream order just before the kernel
    unsigned long long* head = p->d_sample_next + (p->sample_next_slot++ & 63u);
    LTP_HIP_TRY(p, hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
    p->last_kernel = "k_sample";
    ltp::launch_sample(s, first, count, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), offsets,
                       out, f32, capacity, flags, rows, head, blocks, p->dbg_stamps);
    LTP_HIP_TRY(p, hipGetLastError());
    return LTP_OK;
}

int ltp_sample_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                     const unsigned long long* offsets, double* out, unsigned long long capacity, int flags, void* stream)
{
    return sample_batch_any(p, first, count, in, rec, offsets, out, false, capacity, flags, stream);
}

int ltp_sample_batch_f32(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                         const unsigned long long* offsets, float* out, unsigned long long capacity, int flags, void* stream)
{
    return sample_batch_any(p, first, count, in, rec, offsets, out, true, capacity, flags, stream);
}

// the named form of the sampler's policy (include/ltp_hip.h ltp_sample_opts) -> the flag word the kernels' launcher reads
int ltp_sample_batch_ex(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                        const unsigned long long* offsets, void* out, unsigned long long capacity, const ltp_sample_opts* opts, void* stream)
{
    ltp_sample_opts o;
    memset(&o, 0, sizeof o);
    if (opts) {
        // size-versioned: a caller built against an older (shorter) struct leaves the newer fields at their defaults (0)
        if (opts->size < sizeof(unsigned) + sizeof(int)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "ltp_sample_opts.size is not set");
        memcpy(&o, opts, opts->size < sizeof o ? opts->size : sizeof o);
    }
    if ((o.format != LTP_ROWS_F64 && o.format != LTP_ROWS_F32) || (o.stores != LTP_STORES_NONTEMPORAL && o.stores != LTP_STORES_PLAIN) ||
        o.sampler < LTP_SAMPLER_AUTO || o.sampler > LTP_SAMPLER_TABLE || (o.verdict != LTP_VERDICT_KEEP && o.verdict != LTP_VERDICT_SKIP) ||
        o.interleave < 0 || o.interleave > 0xFFFF || (o.dry_run != 0 && o.dry_run != 1))
        return fail(p, LTP_ERR_INVALID_ARGUMENT, "ltp_sample_opts: a field is out of range");
    int flags = (o.stores == LTP_STORES_NONTEMPORAL ? 1 : 0) | (o.dry_run ? 2 : 0) | (o.verdict == LTP_VERDICT_SKIP ? 16 : 0) | (o.interleave << 8);
    switch (o.sampler) {
    case LTP_SAMPLER_FUSED: flags |= 8 | 32; break;
    case LTP_SAMPLER_WALK: flags |= 64; break;
    case LTP_SAMPLER_WALK_STREAMING: flags |= 64 | 128; break;
    case LTP_SAMPLER_TABLE: flags |= 4 | 32; break;
    default: break;
    }
    return sample_batch_any(p, first, count, in, rec, offsets, out, o.format == LTP_ROWS_F32, capacity, flags, stream);
}

int ltp_envelope_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                       int window, int n_windows, double* env, void* stream)
{
    if (!p || first < 0 || count < 0 || !in || !records_complete(rec) || !env) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    if (window < 1 || n_windows < 1) return fail(p, LTP_ERR_INVALID_ARGUMENT, "window and n_windows must be >= 1");
    if (((uintptr_t)env & 15u) != 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "envelope buffer must be 16-byte aligned");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc == LTP_OK) rc = check_geometry(p);
    if (rc != LTP_OK) return rc;
    if (count == 0 || p->dof == 0) return LTP_OK;
    if ((rc = reserve(p, 0)) != LTP_OK) return rc;
    const hipStream_t s = (hipStream_t)stream;
    const int blocks = p->sample_blocks_override > 0 ? p->sample_blocks_override : p->sample_blocks[2];
    // analytic envelopes: the register walk (no tables, no workspace); ltp_set_table_pass(p, 1 | -1) asks for the block-cooperative kernel's
    // analytic form instead (through the table pass / with the build inside the kernel) — same values, A/B runs
    if (p->envelope_mode == LTP_ENVELOPE_ANALYTIC && p->table_pass == 0 && !p->dbg_stamps) {
        ltp::launch_envelope_walk(s, first, count, first, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), window, n_windows, env, p->semantics);
        LTP_HIP_TRY(p, hipGetLastError());
        p->last_kernel = p->semantics == LTP_SEMANTICS_MATLAB ? "k_envelope_walk_matlab analytic" : "k_envelope_walk analytic";
        return LTP_OK;
    }
    if (p->semantics == LTP_SEMANTICS_MATLAB || (!p->dbg_stamps && (p->table_pass > 0 || (p->table_pass == 0 && kEnvelopeTablePassByDefault)))) {
        bool capturing = false;
        if ((rc = workspace_acquire(p, s, capturing)) != LTP_OK) return rc;
        long long piece = 0;
        if ((rc = ensure_tables(p, count, capturing, &piece)) != LTP_OK) return rc;
        for (long long f = first; f < first + count; f += piece) {
            const long long c = first + count - f < piece ? first + count - f : piece;
            unsigned long long* head = p->d_sample_next + (p->sample_next_slot++ & 63u);
            LTP_HIP_TRY(p, hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
            ltp::launch_build_tables(s, f, c, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), ltp::RowSpec{0, 1}, true, nullptr, f, p->d_tables, p->semantics);
            ltp::launch_envelope(s, f, c, first, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), window, n_windows, env, head,
                                 blocks, nullptr, p->d_tables, p->envelope_mode == LTP_ENVELOPE_ANALYTIC);
        }
        LTP_HIP_TRY(p, hipGetLastError());
        p->last_kernel = p->envelope_mode == LTP_ENVELOPE_ANALYTIC ? "k_envelope analytic (run tables from k_build_tables)" : "k_envelope (run tables from k_build_tables)";
        return workspace_release(p, s, capturing);
    }
    unsigned long long* head = p->d_sample_next + (p->sample_next_slot++ & 63u);
    LTP_HIP_TRY(p, hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
    p->last_kernel = p->envelope_mode == LTP_ENVELOPE_ANALYTIC && !p->dbg_stamps ? "k_envelope analytic" : "k_envelope";
    ltp::launch_envelope(s, first, count, first, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), window,
                         n_windows, env, head, blocks, p->dbg_stamps, nullptr, p->envelope_mode == LTP_ENVELOPE_ANALYTIC);
    LTP_HIP_TRY(p, hipGetLastError());
    return LTP_OK;
}

unsigned long long ltp_run_tables_bytes(const ltp_planner* p, long long n_plans)
{
    if (!p || n_plans <= 0 || p->dof <= 0) return 0ull;
    return ltp::table_bytes(n_plans * (long long)p->dof);
}

int ltp_build_tables_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                           unsigned long long* tables, unsigned long long bytes, void* stream)
{
    if (!p || first < 0 || count < 0 || !in || !records_complete(rec) || (!tables && count > 0)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    if (((uintptr_t)tables & 15u) != 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "run-table buffer must be 16-byte aligned");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc == LTP_OK) rc = check_geometry(p);
    if (rc != LTP_OK) return rc;
    if (count == 0 || p->dof == 0) return LTP_OK;
    if (bytes < ltp::table_bytes(count * (long long)p->dof))
        return fail(p, LTP_ERR_INVALID_ARGUMENT, "run-table buffer smaller than ltp_run_tables_bytes(p, count)");
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    // whole tables (every run of every joint), lane = (plan - first) * dof + joint; the caller's buffer, not the handle's workspace
    ltp::launch_build_tables((hipStream_t)stream, first, count, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), ltp::RowSpec{0, 1}, true,
                             nullptr, first, tables, p->semantics);
    LTP_HIP_TRY(p, hipGetLastError());
    return LTP_OK;
}

static int replan_states_any(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                             const unsigned long long* offsets, const void* tile, bool f32, unsigned long long capacity,
                             const int* sample_index, int uniform_index, double* q_0, double* v_0, double* a_0,
                             long long query_stride, long long joint_stride, void* stream)
{
    if (!p || first < 0 || count < 0 || !in || !records_complete(rec) || !offsets || (!tile && capacity > 0) || !q_0 || !v_0 || !a_0)
        return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc == LTP_OK) rc = check_geometry(p);
    if (rc != LTP_OK) return rc;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    ltp::launch_replan_states((hipStream_t)stream, first, count, p->dof, ltp::RowSpec{p->max_samples, p->sample_stride}, to_dev(in), to_dev(rec), offsets, tile, f32,
                              capacity, sample_index, uniform_index, q_0, v_0, a_0, query_stride, joint_stride, p->t_sample, dev_limits(p), p->semantics);
    LTP_HIP_TRY(p, hipGetLastError());
    return LTP_OK;
}

int ltp_state_at_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                       const int* sample_index, int uniform_index, double* q_0, double* v_0, double* a_0,
                       long long query_stride, long long joint_stride, void* stream)
{
    if (!p || first < 0 || count < 0 || !in || !records_complete(rec) || !q_0 || !v_0 || !a_0)
        return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc == LTP_OK) rc = check_geometry(p);
    if (rc != LTP_OK) return rc;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    ltp::launch_state_at((hipStream_t)stream, first, count, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), sample_index,
                         uniform_index, q_0, v_0, a_0, query_stride, joint_stride, p->semantics);
    LTP_HIP_TRY(p, hipGetLastError());
    return LTP_OK;
}

int ltp_replan_states_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                            const unsigned long long* offsets, const double* tile, unsigned long long capacity,
                            const int* sample_index, int uniform_index,
                            double* q_0, double* v_0, double* a_0, long long query_stride, long long joint_stride, void* stream)
{
    return replan_states_any(p, first, count, in, rec, offsets, tile, false, capacity, sample_index, uniform_index, q_0, v_0, a_0,
                             query_stride, joint_stride, stream);
}

int ltp_replan_states_f32_batch(ltp_planner* p, long long first, long long count, const ltp_queries* in, const ltp_records* rec,
                                const unsigned long long* offsets, const float* tile, unsigned long long capacity,
                                const int* sample_index, int uniform_index,
                                double* q_0, double* v_0, double* a_0, long long query_stride, long long joint_stride, void* stream)
{
    return replan_states_any(p, first, count, in, rec, offsets, tile, true, capacity, sample_index, uniform_index, q_0, v_0, a_0,
                             query_stride, joint_stride, stream);
}

int ltp_generate_queries_batch(ltp_planner* p, long long n, unsigned long long seed, long long first_query,
                               double* q_goal, double* q_0, double* v_0, double* a_0,
                               long long query_stride, long long joint_stride, void* stream)
{
    if (!p || n < 0 || !q_goal || !q_0 || !v_0 || !a_0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc != LTP_OK) return rc;
    if (p->dof > 64) return fail(p, LTP_ERR_INVALID_ARGUMENT, "generator supports dof <= 64");
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    ltp::launch_generate((hipStream_t)stream, n, p->dof, dev_limits(p), seed, first_query, q_goal, q_0, v_0, a_0, query_stride,
                         joint_stride);
    LTP_HIP_TRY(p, hipGetLastError());
    return LTP_OK;
}



int ltp_debug_set_sample_stamps(ltp_planner* p, unsigned long long* device_buffer)
{
    if (!p) return LTP_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> g(p->mu);
    p->dbg_stamps = device_buffer;
    return LTP_OK;
}

int ltp_debug_get_sample_blocks(ltp_planner* p, int which)
{
    if (!p || which < 0 || which > 4) return -1;
    std::lock_guard<std::mutex> g(p->mu);
    if (reserve(p, 0) != LTP_OK) return -1;
    return p->sample_blocks[which];
}

int ltp_debug_set_sample_blocks(ltp_planner* p, int blocks)
{
    if (!p || blocks < 0) return LTP_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> g(p->mu);
    p->sample_blocks_override = blocks;
    return LTP_OK;
}

}  // extern "C"
