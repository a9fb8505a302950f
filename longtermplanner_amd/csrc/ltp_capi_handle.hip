// ltp_capi_handle.hip — C ABI (include/ltp_hip.h): handle lifetime, configuration, device workspace.
#include "ltp_handle.hpp"

namespace ltp_capi {

int fail(ltp_planner* p, int code, const std::string& msg)
{
    if (p) p->err = msg;
    return code;
}

int hip_fail(ltp_planner* p, hipError_t e, const char* what)
{
    const int code = (e == hipErrorNoDevice || e == hipErrorInvalidDevice || e == hipErrorInsufficientDriver)
                         ? LTP_ERR_NO_DEVICE
                         : (e == hipErrorOutOfMemory ? LTP_ERR_OUT_OF_MEMORY : LTP_ERR_HIP);
    return fail(p, code, std::string(what) + ": " + hipGetErrorString(e));
}


int upload_limits(ltp_planner* p)
{
    int n = 0;
    for (int k = 0; k < 5; ++k) n = (int)p->h_lim[k].size() > n ? (int)p->h_lim[k].size() : n;
    if (n < 1) n = 1;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    if (n > p->lim_cap) {
        if (p->d_lim) LTP_HIP_TRY(p, hipFree(p->d_lim));
        p->d_lim = nullptr;
        // five limit vectors, then the joints' LimPow tables under LTP_POW_EXACT and under LTP_POW_LIBM (dev_limits picks one)
        LTP_HIP_TRY(p, hipMalloc((void**)&p->d_lim, sizeof(double) * (5 + 2 * ltp::kLimPowN) * n));
        p->lim_cap = n;
    }
    std::vector<double> flat(5 * (size_t)p->lim_cap, 0.0);
    for (int k = 0; k < 5; ++k)
        for (size_t i = 0; i < p->h_lim[k].size(); ++i) flat[(size_t)k * p->lim_cap + i] = p->h_lim[k][i];
    LTP_HIP_TRY(p, hipMemcpy(p->d_lim, flat.data(), sizeof(double) * flat.size(), hipMemcpyHostToDevice));
    // the powers of the limits, by the kernels' own device functions (never inside a capture: ltp_create / ltp_set_limits are not capturable)
    double* pw = p->d_lim + 5 * (size_t)p->lim_cap;
    ltp::launch_limit_powers(nullptr, p->lim_cap, p->d_lim + 3 * (size_t)p->lim_cap, p->d_lim + 4 * (size_t)p->lim_cap, pw, pw + (size_t)ltp::kLimPowN * p->lim_cap);
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
    return LTP_OK;
}

ltp::Limits dev_limits(const ltp_planner* p)
{
    ltp::Limits L;
    L.q_min = p->d_lim;
    L.q_max = p->d_lim + p->lim_cap;
    L.v_max = p->d_lim + 2 * (size_t)p->lim_cap;
    L.a_max = p->d_lim + 3 * (size_t)p->lim_cap;
    L.j_max = p->d_lim + 4 * (size_t)p->lim_cap;
    L.pw = p->d_lim + (5 + (p->pow_rule == LTP_POW_LIBM ? ltp::kLimPowN : 0)) * (size_t)p->lim_cap;
    return L;
}

// the reference indexes its limit vectors unchecked (UB when short); here it is an error
int check_config(ltp_planner* p)
{
    if (p->dof < 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "dof < 0");
    for (int k = 0; k < 5; ++k)
        if ((int)p->h_lim[k].size() < p->dof) return fail(p, LTP_ERR_INVALID_ARGUMENT, "a limit vector has fewer than dof entries");
    if (!(p->t_sample > 0.0)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "t_sample must be > 0");
    return LTP_OK;
}

int reserve(ltp_planner* p, long long n)
{
    const long long items = n * (long long)(p->dof > 0 ? p->dof : 1);
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    if (!p->d_queue_count) LTP_HIP_TRY(p, hipMalloc((void**)&p->d_queue_count, 16 * sizeof(unsigned long long)));
    if (!p->d_small) {
        LTP_HIP_TRY(p, hipMalloc((void**)&p->d_small, sizeof(double) * 16));
        LTP_HIP_TRY(p, hipMemset(p->d_small, 0, sizeof(double) * 16));   // word 0: arrival counter of k_plan_small
    }
    if (!p->d_sample_next) LTP_HIP_TRY(p, hipMalloc((void**)&p->d_sample_next, sizeof(unsigned long long) * 64));
    for (int w = 0; w < 3; ++w)
        if (p->sample_blocks[w] == 0) p->sample_blocks[w] = ltp::sample_resident_blocks(p->device, w);
    for (int w = 0; w < 2; ++w)
        if (p->sample_blocks[3 + w] == 0) p->sample_blocks[3 + w] = ltp::sample_tab_resident_blocks(p->device, w == 1);
    if (p->walk_auto_cus == 0) {
        hipError_t e = hipSuccess;
        p->walk_auto_cus = ltp::sample_walk_auto_prepare(p->device, &e);
        LTP_HIP_TRY(p, e);
    }
    const long long queue_entries = 16 * ltp::queue_segment(n, p->dof > 0 ? p->dof : 1);
    if (queue_entries > p->ws_queue_entries) {
        if (p->d_queue) LTP_HIP_TRY(p, hipFree(p->d_queue));
        p->d_queue = nullptr;
        p->ws_queue_entries = 0;
        LTP_HIP_TRY(p, hipMalloc((void**)&p->d_queue, sizeof(unsigned long long) * (size_t)queue_entries));
        p->ws_queue_entries = queue_entries;
    }
    if (items > p->ws_items) {
        if (p->d_lane_flags) LTP_HIP_TRY(p, hipFree(p->d_lane_flags));
        p->d_lane_flags = nullptr;
        p->ws_items = 0;
        LTP_HIP_TRY(p, hipMalloc((void**)&p->d_lane_flags, (size_t)items));
        p->ws_items = items;
    }
    if (n > p->ws_queries) {
        if (p->d_block_sums) LTP_HIP_TRY(p, hipFree(p->d_block_sums));
        if (p->d_offsets_scratch) LTP_HIP_TRY(p, hipFree(p->d_offsets_scratch));
        p->d_block_sums = p->d_offsets_scratch = nullptr;
        const long long nb = (n + ltp::kScanBlock - 1) / ltp::kScanBlock;
        LTP_HIP_TRY(p, hipMalloc((void**)&p->d_block_sums, sizeof(unsigned long long) * (size_t)(nb + 1)));
        LTP_HIP_TRY(p, hipMalloc((void**)&p->d_offsets_scratch, sizeof(unsigned long long) * (size_t)(n + 1)));
        p->ws_queries = n;
    }
    return LTP_OK;
}

ltp::Queries to_dev(const ltp_queries* in)
{
    ltp::Queries q;
    q.q_goal = in->q_goal; q.q_0 = in->q_0; q.v_0 = in->v_0; q.a_0 = in->a_0;
    q.sq = in->query_stride; q.sj = in->joint_stride;
    return q;
}

ltp::Records to_dev(const ltp_records* r)
{
    ltp::Records o;
    o.t_opt = r->t_opt; o.t_scaled = r->t_scaled; o.dir = r->dir; o.v_drive = r->v_drive; o.mod = r->mod;
    o.t_required = r->t_required; o.slowest = r->slowest; o.traj_len = r->traj_len; o.status = r->status;
    return o;
}

bool records_complete(const ltp_records* r)
{
    return r && r->t_opt && r->t_scaled && r->dir && r->v_drive && r->mod && r->t_required && r->slowest && r->traj_len && r->status;
}

// Called with p->mu held, before a call on stream `s` touches the handle's workspace: if the previous user was another
// stream, `s` waits for the event recorded behind that user's work.
int workspace_acquire(ltp_planner* p, hipStream_t s, bool& capturing)
{
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    capturing = false;
    if (s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) capturing = true;
    (void)hipGetLastError();   // legacy-stream queries may leave a sticky "not supported" behind
    if (capturing) return LTP_OK;
    if (!p->ws_event) LTP_HIP_TRY(p, hipEventCreateWithFlags(&p->ws_event, hipEventDisableTiming));
    if (p->ws_used && p->ws_stream != s) LTP_HIP_TRY(p, hipStreamWaitEvent(s, p->ws_event, 0));
    return LTP_OK;
}

int workspace_release(ltp_planner* p, hipStream_t s, bool capturing)
{
    if (capturing) return LTP_OK;
    LTP_HIP_TRY(p, hipEventRecord(p->ws_event, s));
    p->ws_stream = s;
    p->ws_used = true;
    return LTP_OK;
}

void capture_geometry(ltp_planner* p)
{
    p->planned.valid = true;
    p->planned.dof = p->dof;
    p->planned.t_sample = p->t_sample;
    p->planned.max_samples = p->max_samples;
    p->planned.stride = p->sample_stride;
    p->planned.semantics = p->semantics;
}

// consumers of a planned batch: the handle must still have the geometry the batch was planned with
int check_geometry(ltp_planner* p)
{
    const auto& g = p->planned;
    if (g.valid && (g.dof != p->dof || g.t_sample != p->t_sample || g.max_samples != p->max_samples || g.stride != p->sample_stride || g.semantics != p->semantics))
        return fail(p, LTP_ERR_INVALID_ARGUMENT,
                    "dof, t_sample, max_samples, sample_stride or the semantics changed since the batch was planned; plan it again");
    return LTP_OK;
}

// Table pass or fused build? (DESIGN.md "Table pass".) The pass writes and re-reads up to 912 bytes per joint and runs
// the sampler with streaming waves that never wait; the fused build costs every item ~8 us of latency, three barriers and
// a drain of its own stores: the pass pays when a joint's rows are short (measured crossover: a cap between 256 and 512
// float64 samples, and beyond 1024 float32 samples, whose fused kernel only holds 16 waves per CU). `row_bytes` = bytes of one joint's four rows under the cap (0 = no cap).
bool want_table_pass(const ltp_planner* p, unsigned long long row_bytes, bool f32)
{
    if (p->table_pass != 0) return p->table_pass > 0;
    return row_bytes > 0 && row_bytes <= (f32 ? 16384ull : 8192ull);
}

// k_sample_walk_* (tables built inside the sampler's block) or the fused build of k_sample? Measured, 1 M panda plans
// unless noted (profiles/r04_whole_rows_walk_ab.txt, walk vs fused in TB/s): what decides is how many bytes a plan's rows have —
// below ~150 KB the fused sampler's per-plan build shows. First-512 float64 7.16 vs 6.16, every 3rd sample 6.71 vs 5.85, every 4th
// 6.29 vs 4.75, float32 every 4th sample 4.64 vs 2.40, whole float32 rows 6.83 vs 6.59 (S-ref: 6.80 vs 6.88); level or just behind
// from ~190 KB per plan: first-1024 float64 7.02 vs 7.09, every 2nd sample 6.85 vs 6.91, whole float64 rows 7.03 vs 7.08, S-ref every
// 4th sample 6.93 vs 6.97. The lengths are not known on the host, so the rule goes by what is: the cap, the stride, the element type.
bool want_walk(const ltp_planner* p, int max_samples, int stride, bool f32)
{
    if (p->table_pass != 0) return p->table_pass > 0;
    return f32 || stride >= 3 || (max_samples > 0 && max_samples <= 768);
}

// plans per piece so that the tables of a piece fit the workspace; grows the workspace (up to tables_cap) if needed.
// While a stream is being captured into a hipGraph nothing may be allocated or freed (and a graph that was already
// instantiated keeps the old pointer): then the range is cut into pieces that fit the workspace as it is
// (ltp_reserve_tables sizes it ahead of time), and a handle without any table workspace is an error.
int ensure_tables(ltp_planner* p, long long count, bool capturing, long long* plans_per_piece)
{
    const long long dof = p->dof;
    const unsigned long long per_tile = ltp::table_bytes(64);
    unsigned long long want = ltp::table_bytes(count * dof);
    const unsigned long long cap = p->tables_cap < per_tile * (unsigned long long)dof ? per_tile * (unsigned long long)dof : p->tables_cap;
    if (want > cap) want = cap / per_tile * per_tile;
    if (want > p->tables_bytes && capturing) {
        if (p->tables_bytes < per_tile * (unsigned long long)dof)
            return fail(p, LTP_ERR_INVALID_ARGUMENT,
                        "the table pass needs its workspace, which cannot be allocated while the stream is being captured: "
                        "call ltp_reserve_tables before hipStreamBeginCapture");
        want = p->tables_bytes;   // more pieces, same buffer
    }
    if (want > p->tables_bytes) {
        if (p->d_tables) LTP_HIP_TRY(p, hipFree(p->d_tables));
        p->d_tables = nullptr;
        p->tables_bytes = 0;
        // the workspace is a convenience: when the device cannot spare `want` bytes, take what it can (more pieces)
        for (;;) {
            const hipError_t e = hipMalloc((void**)&p->d_tables, (size_t)want);
            if (e == hipSuccess) break;
            p->d_tables = nullptr;
            (void)hipGetLastError();
            if (e != hipErrorOutOfMemory || want / 2 < per_tile * (unsigned long long)dof) LTP_HIP_TRY(p, e);
            want = want / 2 / per_tile * per_tile;
        }
        p->tables_bytes = want;
    }
    long long plans = (long long)(p->tables_bytes / per_tile) * 64 / dof;
    if (plans < 1) return fail(p, LTP_ERR_OUT_OF_MEMORY, "table workspace too small for one plan");
    *plans_per_piece = plans < count ? plans : count;
    return LTP_OK;
}

}  // namespace ltp_capi

using namespace ltp_capi;

extern "C" {

int ltp_create(int dof, double t_sample, const double* q_min, const double* q_max, const double* v_max,
               const double* a_max, const double* j_max, int device, ltp_planner** out)
{
    if (!out) return LTP_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (dof < 0 || (dof > 0 && (!q_min || !q_max || !v_max || !a_max || !j_max))) return LTP_ERR_INVALID_ARGUMENT;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return LTP_ERR_NO_DEVICE;
    if (device < 0 || device >= count) return LTP_ERR_NO_DEVICE;
    ltp_planner* p = new ltp_planner();
    p->dof = dof;
    p->t_sample = t_sample;
    p->device = device;
    const double* src[5] = {q_min, q_max, v_max, a_max, j_max};
    for (int k = 0; k < 5; ++k) p->h_lim[k].assign(src[k], src[k] + dof);
    int rc = upload_limits(p);
    if (rc == LTP_OK) rc = reserve(p, 1);
    if (rc != LTP_OK) { ltp_destroy(p); return rc; }
    size_t mem_free = 0, mem_total = 0;
    if (hipMemGetInfo(&mem_free, &mem_total) == hipSuccess && (unsigned long long)mem_total / 16 > p->tables_cap)
        p->tables_cap = (unsigned long long)mem_total / 16;
    *out = p;
    return LTP_OK;
}

void ltp_destroy(ltp_planner* p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->d_lim) (void)hipFree(p->d_lim);
    if (p->d_queue) (void)hipFree(p->d_queue);
    if (p->d_lane_flags) (void)hipFree(p->d_lane_flags);
    if (p->d_queue_count) (void)hipFree(p->d_queue_count);
    if (p->d_block_sums) (void)hipFree(p->d_block_sums);
    if (p->d_offsets_scratch) (void)hipFree(p->d_offsets_scratch);
    if (p->d_small) (void)hipFree(p->d_small);
    if (p->d_tables) (void)hipFree(p->d_tables);
    if (p->d_sample_next) (void)hipFree(p->d_sample_next);
    if (p->d_arena) (void)hipFree(p->d_arena);
    if (p->h_arena) (void)hipHostFree(p->h_arena);
    if (p->d_traj) (void)hipFree(p->d_traj);
    if (p->h_traj) (void)hipHostFree(p->h_traj);
    if (p->ws_event) (void)hipEventDestroy(p->ws_event);
    delete p;
}

int ltp_set_limits(ltp_planner* p, int n_limits, const double* q_min, const double* q_max, const double* v_max,
                   const double* a_max, const double* j_max)
{
    if (!p || n_limits < 0 || (n_limits > 0 && (!q_min || !q_max || !v_max || !a_max || !j_max))) return fail(p, LTP_ERR_INVALID_ARGUMENT, "bad limits");
    // Lock order of the library: host_mu (the synchronous host-pointer calls and their arena) BEFORE mu (configuration and the
    // device workspace), everywhere. Both are held here: no host-pointer call launches between the synchronisation and the
    // upload, and no batched call enqueues a kernel that would read the limits while they change.
    std::lock_guard<std::mutex> hg(p->host_mu);
    std::lock_guard<std::mutex> g(p->mu);
    const double* src[5] = {q_min, q_max, v_max, a_max, j_max};
    for (int k = 0; k < 5; ++k) p->h_lim[k].assign(src[k], src[k] + n_limits);
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    LTP_HIP_TRY(p, hipDeviceSynchronize());   // limits are read by in-flight kernels
    return upload_limits(p);
}

int ltp_set_sample_time(ltp_planner* p, double t_sample)
{
    if (!p) return LTP_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> g(p->mu);
    p->t_sample = t_sample;
    return LTP_OK;
}

int ltp_set_dof(ltp_planner* p, int dof)
{
    if (!p || dof < 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "dof < 0");
    std::lock_guard<std::mutex> g(p->mu);
    p->dof = dof;
    return LTP_OK;
}

int ltp_set_max_samples(ltp_planner* p, int max_samples)
{
    if (!p || max_samples < 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "max_samples < 0");
    std::lock_guard<std::mutex> g(p->mu);
    p->max_samples = max_samples;
    return LTP_OK;
}
int ltp_get_max_samples(const ltp_planner* p) { return p ? p->max_samples : -1; }
int ltp_set_sample_stride(ltp_planner* p, int stride)
{
    if (!p || stride < 1) return fail(p, LTP_ERR_INVALID_ARGUMENT, "stride < 1");
    std::lock_guard<std::mutex> g(p->mu);
    p->sample_stride = stride;
    return LTP_OK;
}
int ltp_get_sample_stride(const ltp_planner* p) { return p ? p->sample_stride : -1; }
int ltp_set_goal_check(ltp_planner* p, int enabled)
{
    if (!p) return LTP_ERR_INVALID_ARGUMENT;
    std::lock_guard<std::mutex> g(p->mu);
    p->goal_check = enabled ? 1 : 0;
    return LTP_OK;
}
int ltp_get_goal_check(const ltp_planner* p) { return p ? p->goal_check : -1; }
int ltp_set_semantics(ltp_planner* p, int semantics)
{
    if (!p || (semantics != LTP_SEMANTICS_CPP && semantics != LTP_SEMANTICS_MATLAB)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "semantics must be LTP_SEMANTICS_CPP or LTP_SEMANTICS_MATLAB");
    std::lock_guard<std::mutex> g(p->mu);
    p->semantics = semantics;
    return LTP_OK;
}
int ltp_get_semantics(const ltp_planner* p) { return p ? p->semantics : -1; }
int ltp_set_envelope_mode(ltp_planner* p, int mode)
{
    if (!p || (mode != LTP_ENVELOPE_EXHAUSTIVE && mode != LTP_ENVELOPE_ANALYTIC)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "envelope mode must be LTP_ENVELOPE_EXHAUSTIVE or LTP_ENVELOPE_ANALYTIC");
    std::lock_guard<std::mutex> g(p->mu);
    p->envelope_mode = mode;
    return LTP_OK;
}
int ltp_get_envelope_mode(const ltp_planner* p) { return p ? p->envelope_mode : -1; }
int ltp_set_pow_rule(ltp_planner* p, int rule)
{
    if (!p || (rule != LTP_POW_EXACT && rule != LTP_POW_LIBM)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "pow rule must be LTP_POW_EXACT or LTP_POW_LIBM");
    std::lock_guard<std::mutex> g(p->mu);
    p->pow_rule = rule;
    return LTP_OK;
}
int ltp_get_pow_rule(const ltp_planner* p) { return p ? p->pow_rule : -1; }
int ltp_set_table_pass(ltp_planner* p, int mode)
{
    if (!p || mode < -1 || mode > 1) return fail(p, LTP_ERR_INVALID_ARGUMENT, "table pass mode must be -1, 0 or 1");
    std::lock_guard<std::mutex> g(p->mu);
    p->table_pass = mode;
    return LTP_OK;
}
int ltp_get_table_pass(const ltp_planner* p) { return p ? p->table_pass : -2; }
int ltp_set_table_workspace(ltp_planner* p, unsigned long long bytes)
{
    if (!p || bytes == 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "table workspace must be > 0 bytes");
    std::lock_guard<std::mutex> g(p->mu);
    p->tables_cap = bytes;
    return LTP_OK;
}
int ltp_stored_samples(const ltp_planner* p, int traj_len)
{
    if (!p || traj_len <= 0) return 0;
    const int cnt = (traj_len + p->sample_stride - 1) / p->sample_stride;
    return (p->max_samples > 0 && cnt > p->max_samples) ? p->max_samples : cnt;
}

int ltp_get_dof(const ltp_planner* p) { return p ? p->dof : -1; }
double ltp_get_sample_time(const ltp_planner* p) { return p ? p->t_sample : 0.0; }
const char* ltp_last_error(const ltp_planner* p) { return p ? p->err.c_str() : "null planner"; }
const char* ltp_last_sampler_kernel(const ltp_planner* p) { return p ? p->last_kernel : ""; }
int ltp_row_stride(int traj_len)
{
    if (traj_len <= 0) return 0;
    return (traj_len + ltp::kRowAlign - 1) / ltp::kRowAlign * ltp::kRowAlign;
}

int ltp_reserve_batch(ltp_planner* p, long long n)
{
    if (!p || n < 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "n < 0");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc != LTP_OK) return rc;
    return reserve(p, n);
}

int ltp_reserve_tables(ltp_planner* p, long long n)
{
    if (!p || n < 0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "n < 0");
    std::lock_guard<std::mutex> g(p->mu);
    int rc = check_config(p);
    if (rc != LTP_OK) return rc;
    if (n == 0 || p->dof == 0) return LTP_OK;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    long long piece = 0;
    return ensure_tables(p, n, false, &piece);
}

}  // extern "C"
