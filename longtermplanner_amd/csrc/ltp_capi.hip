// ltp_capi.hip — the C ABI of libltp_hip.so (include/ltp_hip.h): handle, workspace, launches.
// Host-side only; every computation is a kernel in ltp_stage_kernels.hip / ltp_sampler.hip / ltp_aux_kernels.hip. There is deliberately no CPU
// implementation behind these entry points: without a HIP device they fail with LTP_ERR_NO_DEVICE.
#include "../../include/ltp_hip.h"
#include "ltp_kernels.hpp"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

struct ltp_planner {
    int dof = 0;
    double t_sample = 0.001;
    int device = 0;
    int max_samples = 0;                   // 0 = store whole trajectories (reference behaviour)
    int sample_stride = 1;                 // store every sample_stride-th sample
    int goal_check = 0;                    // 1 = reject q_goal outside [q_min,q_max] up front (reference: unchecked)
    int sample_blocks[5] = {0, 0, 0, 0, 0};   // resident blocks of k_sample f64 / f32, k_envelope, k_sample_tab f64 / f32 (work-queue grids)
    int sample_blocks_override = 0;        // tuning aid (ltp_debug_set_sample_blocks)
    unsigned long long* d_sample_next = nullptr;   // ring of work-queue heads, one per in-flight sampler launch
    unsigned sample_next_slot = 0;
    std::vector<double> h_lim[5];          // q_min, q_max, v_max, a_max, j_max as given (any length)
    double* d_lim = nullptr;               // 5 * lim_cap doubles
    int lim_cap = 0;
    unsigned long long* d_queue = nullptr; // two compaction queues of (query*dof + joint), 8 shards each
    unsigned long long* d_queue_count = nullptr;   // [16]
    signed char* d_lane_flags = nullptr;   // per (query, joint) status bits of stage 1
    unsigned long long* d_block_sums = nullptr;
    unsigned long long* d_offsets_scratch = nullptr;
    long long ws_items = 0;                // capacity of d_lane_flags in (query, joint) items
    long long ws_queue_entries = 0;        // capacity of d_queue in u64 entries
    long long ws_queries = 0;
    int table_pass = 0;                    // 0 = automatic, 1 = always, -1 = never (ltp_set_table_pass)
    unsigned long long* d_tables = nullptr;   // run tables of the table pass (k_build_tables); part of the workspace
    unsigned long long tables_bytes = 0;      // allocated
    unsigned long long tables_cap = 4ull << 30;   // upper bound for d_tables (ltp_create: 1/16 of the device's memory if that
                                                  // is more — 18 GiB of 288); longer ranges are processed in pieces
    double* d_small = nullptr;             // 16 doubles for the one-lane entry points
    bool small_dirty = false;              // a fused small-batch call failed: k_plan_small's arrival word may be non-zero
    const char* last_kernel = "";          // row / envelope kernel of the latest ltp_sample_batch* / ltp_envelope_batch
    int semantics = 0;                     // LTP_SEMANTICS_CPP (the reference's C++, default) or LTP_SEMANTICS_MATLAB
    int last_matlab_flags = 0;             // MATLAB semantics: flags of the latest one-lane call (1 = complex intermediate, 2 = error)
    unsigned long long* dbg_stamps = nullptr; // diagnostic: per-block start/end stamps of k_sample (caller-owned)
    // persistent buffers of the small synchronous host-pointer calls (no hipMalloc per call)
    std::mutex host_mu;
    unsigned char* d_arena = nullptr;
    unsigned char* h_arena = nullptr;      // pinned mirror of d_arena
    size_t arena_bytes = 0;
    double* d_traj = nullptr;
    double* h_traj = nullptr;              // pinned
    size_t traj_doubles = 0;
    // the workspace above has one user at a time: the stream of the latest ltp_plan_switch_times_batch and an event
    // recorded behind its work; a call on another stream waits for that event first (include/ltp_hip.h, "Streams")
    hipStream_t ws_stream = nullptr;
    hipEvent_t ws_event = nullptr;
    bool ws_used = false;
    // geometry of the batches planned by this handle (include/ltp_hip.h, "Batch geometry")
    struct Geometry { bool valid = false; int dof = 0; double t_sample = 0.0; int max_samples = 0; int stride = 1; int semantics = 0; } planned;
    std::mutex mu;
    std::string err;
};

namespace {

constexpr bool kEnvelopeTablePassByDefault = true;   // measured: see DESIGN.md "Table pass"

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

#define LTP_HIP_TRY(p, expr)                                   \
    do {                                                       \
        hipError_t e_ = (expr);                                \
        if (e_ != hipSuccess) return hip_fail((p), e_, #expr); \
    } while (0)

int upload_limits(ltp_planner* p)
{
    int n = 0;
    for (int k = 0; k < 5; ++k) n = (int)p->h_lim[k].size() > n ? (int)p->h_lim[k].size() : n;
    if (n < 1) n = 1;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    if (n > p->lim_cap) {
        if (p->d_lim) LTP_HIP_TRY(p, hipFree(p->d_lim));
        p->d_lim = nullptr;
        LTP_HIP_TRY(p, hipMalloc((void**)&p->d_lim, sizeof(double) * 5 * n));
        p->lim_cap = n;
    }
    std::vector<double> flat(5 * (size_t)p->lim_cap, 0.0);
    for (int k = 0; k < 5; ++k)
        for (size_t i = 0; i < p->h_lim[k].size(); ++i) flat[(size_t)k * p->lim_cap + i] = p->h_lim[k][i];
    LTP_HIP_TRY(p, hipMemcpy(p->d_lim, flat.data(), sizeof(double) * flat.size(), hipMemcpyHostToDevice));
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

// device-side record arrays owned for the duration of a *_host call
struct DevRecords {
    ltp_records r{};
    std::vector<void*> owned;
    ~DevRecords() { for (void* q : owned) (void)hipFree(q); }
    template <class T> hipError_t alloc(T** out, size_t count)
    {
        void* ptr = nullptr;
        hipError_t e = hipMalloc(&ptr, sizeof(T) * (count ? count : 1));
        if (e == hipSuccess) { owned.push_back(ptr); *out = (T*)ptr; }
        return e;
    }
    hipError_t alloc_all(long long n, int dof)
    {
        const size_t nd = (size_t)n * dof;
        hipError_t e;
        if ((e = alloc(&r.t_opt, nd * 7)) != hipSuccess) return e;
        if ((e = alloc(&r.t_scaled, nd * 7)) != hipSuccess) return e;
        if ((e = alloc(&r.dir, nd)) != hipSuccess) return e;
        if ((e = alloc(&r.v_drive, nd)) != hipSuccess) return e;
        if ((e = alloc(&r.mod, nd)) != hipSuccess) return e;
        if ((e = alloc(&r.t_required, (size_t)n)) != hipSuccess) return e;
        if ((e = alloc(&r.slowest, (size_t)n)) != hipSuccess) return e;
        if ((e = alloc(&r.traj_len, (size_t)n)) != hipSuccess) return e;
        return alloc(&r.status, (size_t)n);
    }
};

}  // namespace

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
    ltp::launch_switch_times(s, n, p->dof, p->t_sample, p->goal_check, L, q, r, p->d_lane_flags, p->d_queue, p->d_queue_count, p->semantics);
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
    // MATLAB semantics: the run tables always come from the table pass (k_build_tables<MATLAB>); the sampler kernels that read
    // tables do not depend on the semantics, the fused build of k_sample exists for the C++ semantics only
    const bool matlab = p->semantics == LTP_SEMANTICS_MATLAB;
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
                                 blocks, nullptr, p->d_tables);
        }
        LTP_HIP_TRY(p, hipGetLastError());
        p->last_kernel = "k_envelope (run tables from k_build_tables)";
        return workspace_release(p, s, capturing);
    }
    unsigned long long* head = p->d_sample_next + (p->sample_next_slot++ & 63u);
    LTP_HIP_TRY(p, hipMemsetAsync(head, 0, sizeof(unsigned long long), s));
    p->last_kernel = "k_envelope";
    ltp::launch_envelope(s, first, count, first, p->dof, p->t_sample, dev_limits(p), to_dev(in), to_dev(rec), window,
                         n_windows, env, head, blocks, p->dbg_stamps);
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
                              capacity, sample_index, uniform_index, q_0, v_0, a_0, query_stride, joint_stride);
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

// ---- host-pointer convenience ------------------------------------------------------------------

static int run_sample_to_host(ltp_planner* p, long long n, const ltp_queries& dq, const ltp_records& dr,
                              unsigned long long* d_offsets, unsigned long long* offsets, double** packed)
{
    LTP_HIP_TRY(p, hipMemcpy(offsets, d_offsets, sizeof(unsigned long long) * (size_t)(n + 1), hipMemcpyDeviceToHost));
    const unsigned long long total = offsets[n];
    double* d_out = nullptr;
    LTP_HIP_TRY(p, hipMalloc((void**)&d_out, sizeof(double) * (size_t)(total ? total : 2)));
    // row padding beyond a row's last 16-byte slot is never written by the sampler (the tail of that slot is
    // zero-filled): make the host copy deterministic
    hipError_t e = hipMemset(d_out, 0, sizeof(double) * (size_t)(total ? total : 2));
    int rc = LTP_OK;
    if (e != hipSuccess) rc = hip_fail(p, e, "hipMemset");
    if (rc == LTP_OK) rc = ltp_sample_batch(p, 0, n, &dq, &dr, d_offsets, d_out, total, 0, nullptr);
    if (rc == LTP_OK) {
        e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess) rc = hip_fail(p, e, "hipStreamSynchronize");
    }
    if (rc == LTP_OK) {
        double* h = (double*)malloc(sizeof(double) * (size_t)(total ? total : 1));
        if (!h) rc = fail(p, LTP_ERR_OUT_OF_MEMORY, "malloc");
        else {
            e = hipMemcpy(h, d_out, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost);
            if (e != hipSuccess) { free(h); rc = hip_fail(p, e, "hipMemcpy"); }
            else *packed = h;
        }
    }
    (void)hipFree(d_out);
    return rc;
}

static int download_records(ltp_planner* p, long long n, int dof, const ltp_records& d, const ltp_records* h)
{
    if (!h) return LTP_OK;
    const size_t nd = (size_t)n * dof;
    if (h->t_opt) LTP_HIP_TRY(p, hipMemcpy(h->t_opt, d.t_opt, sizeof(double) * nd * 7, hipMemcpyDeviceToHost));
    if (h->t_scaled) LTP_HIP_TRY(p, hipMemcpy(h->t_scaled, d.t_scaled, sizeof(double) * nd * 7, hipMemcpyDeviceToHost));
    if (h->dir) LTP_HIP_TRY(p, hipMemcpy(h->dir, d.dir, sizeof(double) * nd, hipMemcpyDeviceToHost));
    if (h->v_drive) LTP_HIP_TRY(p, hipMemcpy(h->v_drive, d.v_drive, sizeof(double) * nd, hipMemcpyDeviceToHost));
    if (h->mod) LTP_HIP_TRY(p, hipMemcpy(h->mod, d.mod, nd, hipMemcpyDeviceToHost));
    if (h->t_required) LTP_HIP_TRY(p, hipMemcpy(h->t_required, d.t_required, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    if (h->slowest) LTP_HIP_TRY(p, hipMemcpy(h->slowest, d.slowest, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    if (h->traj_len) LTP_HIP_TRY(p, hipMemcpy(h->traj_len, d.traj_len, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    if (h->status) LTP_HIP_TRY(p, hipMemcpy(h->status, d.status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    return LTP_OK;
}

// ---- pinned result buffers: what ltp_plan_batch_host / ltp_get_trajectory_host hand out as *packed for small batches.
// The fused small-batch kernel writes the rows straight into such a buffer (host memory the device can address), so the
// caller gets them without any copy; ltp_free_host returns the buffer here instead of to the heap. ----
namespace {

struct PinnedPool {
    struct Buf { void* ptr; size_t bytes; bool used; };
    std::mutex mu;
    std::vector<Buf> bufs;
    static constexpr size_t kKeep = 16;           // buffers kept for reuse

    void* acquire(size_t bytes)
    {
        std::lock_guard<std::mutex> g(mu);
        for (auto& b : bufs)
            if (!b.used && b.bytes >= bytes) { b.used = true; return b.ptr; }
        void* ptr = nullptr;
        // the pool is process-wide and its buffers are handed to kernels on any device (ltp_plan_batch_multi): portable, and
        // explicitly coherent — Portable alone makes the memory non-coherent, and the completion word the host spins on
        // (wait_done) as well as the rows themselves rely on coherence
        if (hipHostMalloc(&ptr, bytes, hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        bufs.push_back(Buf{ptr, bytes, true});
        return ptr;
    }
    // true if ptr is one of ours
    bool release(void* ptr)
    {
        std::lock_guard<std::mutex> g(mu);
        size_t idle = 0;
        for (auto& b : bufs) idle += !b.used;
        for (size_t i = 0; i < bufs.size(); ++i)
            if (bufs[i].ptr == ptr) {
                if (idle >= kKeep) { (void)hipHostFree(ptr); bufs.erase(bufs.begin() + (long)i); }
                else bufs[i].used = false;
                return true;
            }
        return false;
    }
};
PinnedPool g_pinned;

constexpr size_t kFusedRowsBytes = 8u << 20;      // rows of a fused small-batch call: up to 1 Mi doubles (7-DoF, 1 ms: 48 k)

// waits for the kernel's completion word in pinned memory (a few microseconds sooner than a stream synchronisation)
int wait_done(ltp_planner* p, volatile int* done)
{
    for (long spins = 0; *done == 0; ++spins) {
        if (spins > 2000000) {                      // ~ a second without news: ask the runtime (reports a faulted kernel)
            LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
            if (*done == 0) return fail(p, LTP_ERR_HIP, "small-batch kernel finished without reporting");
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    std::atomic_thread_fence(std::memory_order_acquire);   // the result buffers are read after the flag
    return LTP_OK;
}

}  // namespace

// ---- small-batch host path: one persistent device arena + pinned mirror, one H2D and one D2H per call ----
namespace {

constexpr size_t kSmallHostBytes = 8u << 20;   // batches whose arena fits in 8 MiB take the staged path

struct ArenaLayout {
    size_t in[4], t_opt, t_scaled, dir, v_drive, t_required, offsets, slowest, traj_len, status, mod, end, rec_begin;
};

ArenaLayout arena_layout(long long n, int dof)
{
    const size_t nd = (size_t)n * dof;
    ArenaLayout L;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 15) & ~(size_t)15; return at; };
    for (int k = 0; k < 4; ++k) L.in[k] = take(sizeof(double) * nd);
    L.rec_begin = o;
    L.t_opt = take(sizeof(double) * nd * 7);
    L.t_scaled = take(sizeof(double) * nd * 7);
    L.dir = take(sizeof(double) * nd);
    L.v_drive = take(sizeof(double) * nd);
    L.t_required = take(sizeof(double) * (size_t)n);
    L.offsets = take(sizeof(unsigned long long) * ((size_t)n + 1));
    L.slowest = take(sizeof(int) * (size_t)n);
    L.traj_len = take(sizeof(int) * (size_t)n);
    L.status = take(sizeof(int) * (size_t)n);
    L.mod = take(nd);
    L.end = o;
    return L;
}

int ensure_arena(ltp_planner* p, size_t bytes)
{
    if (bytes <= p->arena_bytes) return LTP_OK;
    if (p->d_arena) LTP_HIP_TRY(p, hipFree(p->d_arena));
    if (p->h_arena) LTP_HIP_TRY(p, hipHostFree(p->h_arena));
    p->d_arena = nullptr; p->h_arena = nullptr; p->arena_bytes = 0;
    const size_t cap = bytes < 65536 ? 65536 : bytes;
    LTP_HIP_TRY(p, hipMalloc((void**)&p->d_arena, cap));
    LTP_HIP_TRY(p, hipHostMalloc((void**)&p->h_arena, cap, hipHostMallocDefault));
    p->arena_bytes = cap;
    return LTP_OK;
}

constexpr size_t kPinnedTrajDoubles = (32u << 20) / sizeof(double);   // pinned staging only for small results

int ensure_traj(ltp_planner* p, size_t doubles)
{
    if (doubles <= p->traj_doubles) return LTP_OK;
    if (p->d_traj) LTP_HIP_TRY(p, hipFree(p->d_traj));
    if (p->h_traj) LTP_HIP_TRY(p, hipHostFree(p->h_traj));
    p->d_traj = nullptr; p->h_traj = nullptr; p->traj_doubles = 0;
    LTP_HIP_TRY(p, hipMalloc((void**)&p->d_traj, sizeof(double) * doubles));
    if (doubles <= kPinnedTrajDoubles) LTP_HIP_TRY(p, hipHostMalloc((void**)&p->h_traj, sizeof(double) * doubles, hipHostMallocDefault));
    p->traj_doubles = doubles;
    return LTP_OK;
}

ltp_records arena_records(unsigned char* base, const ArenaLayout& L)
{
    ltp_records r;
    r.t_opt = (double*)(base + L.t_opt); r.t_scaled = (double*)(base + L.t_scaled); r.dir = (double*)(base + L.dir);
    r.v_drive = (double*)(base + L.v_drive); r.mod = (signed char*)(base + L.mod); r.t_required = (double*)(base + L.t_required);
    r.slowest = (int*)(base + L.slowest); r.traj_len = (int*)(base + L.traj_len); r.status = (int*)(base + L.status);
    return r;
}

// sample all n plans of an arena batch into the cached device buffer and hand back a malloc'ed host copy;
// also refreshes the arena's host copy of `status` (the sampler may set LTP_STATUS_END_LIMIT)
int sample_to_host_small(ltp_planner* p, long long n, const ArenaLayout& L, const ltp_queries& dq, const ltp_records& dr,
                         unsigned long long* d_off, unsigned long long total, double** packed)
{
    int rc = ensure_traj(p, (size_t)(total ? total : 2));
    if (rc != LTP_OK) return rc;
    // row padding beyond a row's last 16-byte slot is never written by the sampler (the tail of that slot is
    // zero-filled): make the host copy deterministic
    LTP_HIP_TRY(p, hipMemsetAsync(p->d_traj, 0, sizeof(double) * (size_t)(total ? total : 2), nullptr));
    rc = ltp_sample_batch(p, 0, n, &dq, &dr, d_off, p->d_traj, total, 0, nullptr);
    if (rc != LTP_OK) return rc;
    double* h = (double*)malloc(sizeof(double) * (size_t)(total ? total : 1));
    if (!h) return fail(p, LTP_ERR_OUT_OF_MEMORY, "malloc");
    double* landing = p->h_traj ? p->h_traj : h;   // pinned staging when the result is small
    hipError_t e = total ? hipMemcpyAsync(landing, p->d_traj, sizeof(double) * (size_t)total, hipMemcpyDeviceToHost, nullptr) : hipSuccess;
    if (e == hipSuccess) e = hipMemcpyAsync(p->h_arena + L.status, p->d_arena + L.status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, nullptr);
    if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
    if (e != hipSuccess) { free(h); return hip_fail(p, e, "trajectory download"); }
    if (total && landing != h) memcpy(h, landing, sizeof(double) * (size_t)total);
    *packed = h;
    return LTP_OK;
}

// The fused path of ltp_plan_batch_host / ltp_get_trajectory_host for n * dof <= small_batch_pairs(): one launch of one
// block that reads the queries from and writes records and rows to pinned host memory (k_plan_small), one wait. Caller holds
// host_mu. Returns LTP_OK with *handled = false when the rows do not fit the pinned result buffer (caller takes the staged path).
int plan_batch_host_fused(ltp_planner* p, long long n, const double* const (&h_in)[4], const ltp_records* host_records,
                          const ltp_records* given /* getTrajectory: t_scaled, dir, mod, v_drive are inputs */,
                          unsigned long long* offsets, double** packed, bool* handled)
{
    *handled = false;
    const int dof = p->dof;
    const size_t nd = (size_t)n * dof;
    const ArenaLayout L = arena_layout(n, dof);
    const size_t flag_at = (L.end + 63) & ~(size_t)63;
    const int blocks = ltp::small_batch_blocks(dof, packed != nullptr);
    const size_t ends_at = flag_at + 64;                         // [blocks][n] end-limit bits
    int rc = ensure_arena(p, ends_at + sizeof(int) * (size_t)blocks * (size_t)n);
    if (rc != LTP_OK) return rc;
    for (int k = 0; k < 4; ++k)
        if (h_in[k]) memcpy(p->h_arena + L.in[k], h_in[k], sizeof(double) * nd);
    const ltp_records hr = arena_records(p->h_arena, L);
    if (given) {
        memcpy(hr.t_scaled, given->t_scaled, sizeof(double) * nd * 7);
        memcpy(hr.dir, given->dir, sizeof(double) * nd);
        memcpy(hr.v_drive, given->v_drive, sizeof(double) * nd);
        memcpy(hr.mod, given->mod, nd);
    }
    double* rows = nullptr;
    if (packed) {
        rows = (double*)g_pinned.acquire(kFusedRowsBytes);
        if (!rows) return LTP_OK;                                // no pinned memory to be had: staged path
    }
    if (p->small_dirty) {
        // an earlier fused call failed or was abandoned: whatever it left running must be over and k_plan_small's arrival
        // word zero again before the next launch counts on it
        LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
        LTP_HIP_TRY(p, hipMemset(p->d_small, 0, sizeof(unsigned int)));
        p->small_dirty = false;
    }
    volatile int* done = (volatile int*)(p->h_arena + flag_at);
    *done = 0;
    const double* in[4] = {(const double*)(p->h_arena + L.in[0]), (const double*)(p->h_arena + L.in[1]),
                           (const double*)(p->h_arena + L.in[2]), (const double*)(p->h_arena + L.in[3])};
    {
        std::lock_guard<std::mutex> g(p->mu);
        capture_geometry(p);
        ltp::launch_plan_small(nullptr, (int)n, dof, p->t_sample, p->goal_check, ltp::RowSpec{p->max_samples, p->sample_stride}, dev_limits(p), in,
                               to_dev(&hr), (unsigned long long*)(p->h_arena + L.offsets), rows, kFusedRowsBytes / sizeof(double),
                               (int*)(p->h_arena + ends_at), (unsigned int*)p->d_small, done, given != nullptr);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { if (rows) g_pinned.release(rows); return hip_fail(p, e, "k_plan_small"); }   // nothing was launched
    }
    rc = wait_done(p, done);
    if (rc != LTP_OK) {
        // the kernel may still be running (or have died half way): its arrival word is suspect, and `rows` goes back to the
        // pool only once the stream is known to be idle — otherwise it stays allocated (leaked) rather than be written behind a later owner's back
        p->small_dirty = true;
        if (rows && hipStreamSynchronize(nullptr) == hipSuccess) g_pinned.release(rows);
        (void)hipGetLastError();
        return rc;
    }
    if (*done == 2) {                                            // rows larger than the pinned buffer
        g_pinned.release(rows);
        return LTP_OK;
    }
    if (packed) {                                                // end-limit bits of the blocks that sampled (cc:59-61)
        const int* ends = (const int*)(p->h_arena + ends_at);
        for (int b = 0; b < blocks; ++b)
            for (long long i = 0; i < n; ++i) hr.status[i] |= ends[(size_t)b * n + i];
    }
    const unsigned long long* h_off = (const unsigned long long*)(p->h_arena + L.offsets);
    if (offsets) memcpy(offsets, h_off, sizeof(unsigned long long) * ((size_t)n + 1));
    if (host_records) {
        if (host_records->t_opt && !given) memcpy(host_records->t_opt, hr.t_opt, sizeof(double) * nd * 7);
        if (host_records->t_scaled && !given) memcpy(host_records->t_scaled, hr.t_scaled, sizeof(double) * nd * 7);
        if (host_records->dir && !given) memcpy(host_records->dir, hr.dir, sizeof(double) * nd);
        if (host_records->v_drive && !given) memcpy(host_records->v_drive, hr.v_drive, sizeof(double) * nd);
        if (host_records->mod && !given) memcpy(host_records->mod, hr.mod, nd);
        if (host_records->t_required && !given) memcpy(host_records->t_required, hr.t_required, sizeof(double) * (size_t)n);
        if (host_records->slowest && !given) memcpy(host_records->slowest, hr.slowest, sizeof(int) * (size_t)n);
        if (host_records->traj_len) memcpy(host_records->traj_len, hr.traj_len, sizeof(int) * (size_t)n);
        if (host_records->status) memcpy(host_records->status, hr.status, sizeof(int) * (size_t)n);
    }
    if (packed) *packed = rows;
    *handled = true;
    return LTP_OK;
}

// the staged path of ltp_plan_batch_host; caller holds host_mu
int plan_batch_host_small(ltp_planner* p, long long n, const double* const (&h_in)[4], const ltp_records* host_records,
                          unsigned long long* offsets, double** packed)
{
    const int dof = p->dof;
    const size_t nd = (size_t)n * dof;
    const ArenaLayout L = arena_layout(n, dof);
    int rc = ensure_arena(p, L.end);
    if (rc != LTP_OK) return rc;
    for (int k = 0; k < 4; ++k)
        if (nd) memcpy(p->h_arena + L.in[k], h_in[k], sizeof(double) * nd);
    if (L.rec_begin) LTP_HIP_TRY(p, hipMemcpyAsync(p->d_arena, p->h_arena, L.rec_begin, hipMemcpyHostToDevice, nullptr));
    const ltp_queries dq{(double*)(p->d_arena + L.in[0]), (double*)(p->d_arena + L.in[1]), (double*)(p->d_arena + L.in[2]),
                         (double*)(p->d_arena + L.in[3]), dof, 1};
    const ltp_records dr = arena_records(p->d_arena, L);
    unsigned long long* d_off = (unsigned long long*)(p->d_arena + L.offsets);
    rc = ltp_plan_switch_times_batch(p, n, &dq, &dr, d_off, nullptr);
    if (rc == LTP_OK && !packed) rc = ltp_end_limit_batch(p, 0, n, &dq, &dr, nullptr);   // cc:59-61 without the sampler
    if (rc != LTP_OK) return rc;
    LTP_HIP_TRY(p, hipMemcpyAsync(p->h_arena + L.rec_begin, p->d_arena + L.rec_begin, L.end - L.rec_begin, hipMemcpyDeviceToHost, nullptr));
    LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
    const unsigned long long* h_off = (const unsigned long long*)(p->h_arena + L.offsets);
    if (packed) {
        rc = sample_to_host_small(p, n, L, dq, dr, d_off, h_off[n], packed);
        if (rc != LTP_OK) return rc;
    }
    if (offsets) memcpy(offsets, h_off, sizeof(unsigned long long) * ((size_t)n + 1));
    if (host_records) {
        const ltp_records hr = arena_records(p->h_arena, L);
        if (host_records->t_opt) memcpy(host_records->t_opt, hr.t_opt, sizeof(double) * nd * 7);
        if (host_records->t_scaled) memcpy(host_records->t_scaled, hr.t_scaled, sizeof(double) * nd * 7);
        if (host_records->dir) memcpy(host_records->dir, hr.dir, sizeof(double) * nd);
        if (host_records->v_drive) memcpy(host_records->v_drive, hr.v_drive, sizeof(double) * nd);
        if (host_records->mod) memcpy(host_records->mod, hr.mod, nd);
        if (host_records->t_required) memcpy(host_records->t_required, hr.t_required, sizeof(double) * (size_t)n);
        if (host_records->slowest) memcpy(host_records->slowest, hr.slowest, sizeof(int) * (size_t)n);
        if (host_records->traj_len) memcpy(host_records->traj_len, hr.traj_len, sizeof(int) * (size_t)n);
        if (host_records->status) memcpy(host_records->status, hr.status, sizeof(int) * (size_t)n);
    }
    return LTP_OK;
}

// one-lane entry points: the kernel reads its 16 doubles from, and writes them back to, the pinned arena (host memory the
// device addresses directly): one launch, one synchronisation, no copy engine
extern "C++" {
template <class Launch>
int run_one_lane(ltp_planner* p, int joint, double (&buf)[16], Launch launch)
{
    std::lock_guard<std::mutex> hg(p->host_mu);        // host_mu before mu (see ltp_set_limits)
    double t_sample;
    int semantics;
    ltp::Limits lim;
    {
        std::lock_guard<std::mutex> g(p->mu);
        if (joint < 0 || joint >= p->lim_cap) return fail(p, LTP_ERR_INVALID_ARGUMENT, "joint out of range");
        t_sample = p->t_sample;
        semantics = p->semantics;
        lim = dev_limits(p);                           // stays valid: ltp_set_limits needs host_mu, which this call holds
    }
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    int rc = ensure_arena(p, sizeof(buf));
    if (rc != LTP_OK) return rc;
    memcpy(p->h_arena, buf, sizeof(buf));
    launch((double*)p->h_arena, t_sample, lim, semantics);
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
    memcpy(buf, p->h_arena, sizeof(buf));
    p->last_matlab_flags = (int)buf[11];
    return LTP_OK;
}
}  // extern "C++"

}  // namespace

int ltp_plan_batch_host(ltp_planner* p, long long n, const double* q_goal, const double* q_0, const double* v_0,
                        const double* a_0, const ltp_records* host_records, unsigned long long* offsets, double** packed)
{
    if (!p || n < 0 || (packed && !offsets)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    if (n > 0 && p->dof > 0 && (!q_goal || !q_0 || !v_0 || !a_0)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null query array");
    if (packed) *packed = nullptr;
    int rc;
    { std::lock_guard<std::mutex> g(p->mu); rc = check_config(p); }
    if (rc != LTP_OK) return rc;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    const int dof = p->dof;
    const size_t nd = (size_t)n * dof;
    const double* const h_in[4] = {q_goal, q_0, v_0, a_0};
    if (n > 0 && dof > 0 && arena_layout(n, dof).end <= kSmallHostBytes) {
        std::lock_guard<std::mutex> hg(p->host_mu);
        if (nd <= (size_t)ltp::small_batch_pairs() && p->semantics == LTP_SEMANTICS_CPP) {   // k_plan_small exists for the C++ semantics only
            bool handled = false;
            rc = plan_batch_host_fused(p, n, h_in, host_records, nullptr, offsets, packed, &handled);
            if (rc != LTP_OK || handled) return rc;
        }
        return plan_batch_host_small(p, n, h_in, host_records, offsets, packed);
    }
    DevRecords dr;
    LTP_HIP_TRY(p, dr.alloc_all(n, dof));
    double* d_in[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; ++k) {
        LTP_HIP_TRY(p, dr.alloc(&d_in[k], nd));
        if (nd) LTP_HIP_TRY(p, hipMemcpy(d_in[k], h_in[k], sizeof(double) * nd, hipMemcpyHostToDevice));
    }
    unsigned long long* d_off = nullptr;
    LTP_HIP_TRY(p, dr.alloc(&d_off, (size_t)n + 1));
    ltp_queries dq{d_in[0], d_in[1], d_in[2], d_in[3], dof, 1};
    rc = ltp_plan_switch_times_batch(p, n, &dq, &dr.r, d_off, nullptr);
    if (rc == LTP_OK && !packed) rc = ltp_end_limit_batch(p, 0, n, &dq, &dr.r, nullptr);   // cc:59-61 without the sampler
    if (rc != LTP_OK) return rc;
    LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
    if (packed) {
        rc = run_sample_to_host(p, n, dq, dr.r, d_off, offsets, packed);
        if (rc != LTP_OK) return rc;
    } else if (offsets) {
        LTP_HIP_TRY(p, hipMemcpy(offsets, d_off, sizeof(unsigned long long) * (size_t)(n + 1), hipMemcpyDeviceToHost));
    }
    return download_records(p, n, dof, dr.r, host_records);   // after sampling: status carries END_LIMIT
}

void ltp_shard_range(long long n, int rank, int world, long long* first, long long* count)
{
    long long f = 0, c = 0;
    if (n > 0 && world > 0 && rank >= 0 && rank < world) {
        const long long base = n / world, rem = n % world;
        c = base + (rank < rem ? 1 : 0);
        f = rank * base + (rank < rem ? rank : rem);
    }
    if (first) *first = f;
    if (count) *count = c;
}

namespace {

// all planners of a *_multi call must be distinct handles configured like planners[0]; the error text lands in planners[0]
int check_shard_planners(ltp_planner* const* planners, int k)
{
    ltp_planner* p0 = planners[0];
    for (int g = 1; g < k; ++g) {
        const ltp_planner* pg = planners[g];
        bool same = pg && pg->dof == p0->dof && pg->t_sample == p0->t_sample && pg->max_samples == p0->max_samples &&
                    pg->sample_stride == p0->sample_stride && pg->goal_check == p0->goal_check && pg->semantics == p0->semantics;
        for (int l = 0; same && l < 5; ++l) {
            same = (int)pg->h_lim[l].size() >= p0->dof && (int)p0->h_lim[l].size() >= p0->dof;
            for (int j = 0; same && j < p0->dof; ++j) same = pg->h_lim[l][j] == p0->h_lim[l][j];
        }
        if (!same) return fail(p0, LTP_ERR_INVALID_ARGUMENT, "planner " + std::to_string(g) + " is not configured like planner 0");
        for (int h = 0; h < g; ++h)
            if (planners[h] == pg) return fail(p0, LTP_ERR_INVALID_ARGUMENT, "the same planner handle is listed twice");
    }
    return LTP_OK;
}

// run(g) for every shard, one host thread per shard (each binds its own device); a thread that cannot be created
// (std::system_error must not cross the C boundary) runs its shard inline instead. Returns the first failing shard's code.
extern "C++" {
template <class Run>
int run_shards(ltp_planner* const* planners, int k, Run run)
{
    std::vector<int> rcs((size_t)k, LTP_OK);
    {
        std::vector<std::thread> th;
        th.reserve((size_t)k);
        for (int g = 1; g < k; ++g) {
            try {
                th.emplace_back([&rcs, &run, g] { rcs[(size_t)g] = run(g); });
            } catch (const std::system_error&) {
                rcs[(size_t)g] = run(g);
            }
        }
        rcs[0] = run(0);
        for (auto& t : th) t.join();
    }
    for (int g = 0; g < k; ++g)
        if (rcs[(size_t)g] != LTP_OK)
            return fail(planners[0], rcs[(size_t)g],
                        "shard " + std::to_string(g) + " (device " + std::to_string(planners[g]->device) + "): " + (g ? planners[g]->err : std::string(planners[0]->err)));
    return LTP_OK;
}
}  // extern "C++"

}  // namespace

int ltp_plan_batch_multi(ltp_planner* const* planners, int k, long long n, const double* q_goal, const double* q_0,
                         const double* v_0, const double* a_0, const ltp_records* host_records, unsigned long long* offsets,
                         double** packed)
{
    if (!planners || k < 1 || !planners[0]) return LTP_ERR_INVALID_ARGUMENT;
    ltp_planner* p0 = planners[0];
    if (n < 0 || (packed && !offsets)) return fail(p0, LTP_ERR_INVALID_ARGUMENT, "null argument");
    if (packed) *packed = nullptr;
    int rc = check_shard_planners(planners, k);
    if (rc != LTP_OK) return rc;
    const int dof = p0->dof;
    std::vector<long long> first((size_t)k), count((size_t)k);
    for (int g = 0; g < k; ++g) ltp_shard_range(n, g, k, &first[g], &count[g]);
    std::vector<double*> parts((size_t)k, nullptr);
    std::vector<std::vector<unsigned long long>> offs((size_t)k);
    auto run = [&](int g) -> int {
        const long long f = first[g], c = count[g];
        const size_t fd = (size_t)f * dof;
        ltp_records r{};
        if (host_records) {
            r = *host_records;
            if (r.t_opt) r.t_opt += fd * 7;
            if (r.t_scaled) r.t_scaled += fd * 7;
            if (r.dir) r.dir += fd;
            if (r.v_drive) r.v_drive += fd;
            if (r.mod) r.mod += fd;
            if (r.t_required) r.t_required += f;
            if (r.slowest) r.slowest += f;
            if (r.traj_len) r.traj_len += f;
            if (r.status) r.status += f;
        }
        offs[g].assign((size_t)c + 1, 0ull);
        return ltp_plan_batch_host(planners[g], c, q_goal ? q_goal + fd : nullptr, q_0 ? q_0 + fd : nullptr, v_0 ? v_0 + fd : nullptr,
                                   a_0 ? a_0 + fd : nullptr, host_records ? &r : nullptr, offsets ? offs[g].data() : nullptr,
                                   packed ? &parts[g] : nullptr);
    };
    rc = run_shards(planners, k, run);
    if (rc == LTP_OK && offsets) {
        unsigned long long base = 0ull;
        for (int g = 0; g < k; ++g) {
            for (long long i = 0; i < count[g]; ++i) offsets[first[g] + i] = base + offs[g][(size_t)i];
            base += offs[g][(size_t)count[g]];
        }
        offsets[n] = base;
        if (packed) {
            double* all = (double*)malloc(sizeof(double) * (size_t)(base ? base : 1));
            if (!all) rc = fail(p0, LTP_ERR_OUT_OF_MEMORY, "malloc");
            else {
                unsigned long long at = 0ull;
                for (int g = 0; g < k; ++g) {
                    const unsigned long long sz = offs[g][(size_t)count[g]];
                    if (sz) memcpy(all + at, parts[g], sizeof(double) * (size_t)sz);
                    at += sz;
                }
                *packed = all;
            }
        }
    }
    for (int g = 0; g < k; ++g) ltp_free_host(parts[g]);   // small shards come from the pinned result pool
    return rc;
}

// ---- device-resident shards: per-shard device pointers, nothing passes through the host ----
namespace {
int check_shards(ltp_planner* const* planners, int k, long long n, const ltp_shard* shards)
{
    if (!planners || k < 1 || !planners[0]) return LTP_ERR_INVALID_ARGUMENT;
    if (n < 0 || !shards) return fail(planners[0], LTP_ERR_INVALID_ARGUMENT, "null argument");
    return check_shard_planners(planners, k);
}
}  // namespace

int ltp_plan_switch_times_multi(ltp_planner* const* planners, int k, long long n, const ltp_shard* shards, int end_limit)
{
    int rc = check_shards(planners, k, n, shards);
    if (rc != LTP_OK) return rc;
    return run_shards(planners, k, [&](int g) -> int {
        long long f = 0, c = 0;
        ltp_shard_range(n, g, k, &f, &c);
        if (c == 0) {
            // an empty tail shard (more shards than queries): its offsets array is the single entry 0
            if (shards[g].offsets) {
                LTP_HIP_TRY(planners[g], hipSetDevice(planners[g]->device));
                LTP_HIP_TRY(planners[g], hipMemsetAsync(shards[g].offsets, 0, sizeof(unsigned long long), (hipStream_t)shards[g].stream));
            }
            return LTP_OK;
        }
        int r = ltp_plan_switch_times_batch(planners[g], c, &shards[g].in, &shards[g].out, shards[g].offsets, shards[g].stream);
        if (r == LTP_OK && end_limit) r = ltp_end_limit_batch(planners[g], 0, c, &shards[g].in, &shards[g].out, shards[g].stream);
        return r;
    });
}

int ltp_envelope_multi(ltp_planner* const* planners, int k, long long n, const ltp_shard* shards, int window, int n_windows,
                       double* const* env)
{
    int rc = check_shards(planners, k, n, shards);
    if (rc != LTP_OK) return rc;
    if (!env) return fail(planners[0], LTP_ERR_INVALID_ARGUMENT, "null argument");
    return run_shards(planners, k, [&](int g) -> int {
        long long f = 0, c = 0;
        ltp_shard_range(n, g, k, &f, &c);
        if (c == 0) return LTP_OK;
        return ltp_envelope_batch(planners[g], 0, c, &shards[g].in, &shards[g].out, window, n_windows, env[g], shards[g].stream);
    });
}

int ltp_state_at_multi(ltp_planner* const* planners, int k, long long n, const ltp_shard* shards, const int* const* sample_index,
                       int uniform_index, double* const* q_0, double* const* v_0, double* const* a_0)
{
    int rc = check_shards(planners, k, n, shards);
    if (rc != LTP_OK) return rc;
    if (!q_0 || !v_0 || !a_0) return fail(planners[0], LTP_ERR_INVALID_ARGUMENT, "null argument");
    return run_shards(planners, k, [&](int g) -> int {
        long long f = 0, c = 0;
        ltp_shard_range(n, g, k, &f, &c);
        if (c == 0) return LTP_OK;
        // the states are laid out like the shard's queries: they are the next batch's q_0, v_0, a_0
        return ltp_state_at_batch(planners[g], 0, c, &shards[g].in, &shards[g].out, sample_index ? sample_index[g] : nullptr,
                                  uniform_index, q_0[g], v_0[g], a_0[g], shards[g].in.query_stride, shards[g].in.joint_stride, shards[g].stream);
    });
}

int ltp_synchronize_multi(ltp_planner* const* planners, int k, const ltp_shard* shards)
{
    if (!planners || k < 1 || !planners[0]) return LTP_ERR_INVALID_ARGUMENT;
    if (!shards) return fail(planners[0], LTP_ERR_INVALID_ARGUMENT, "null argument");
    for (int g = 0; g < k; ++g) {
        ltp_planner* p = planners[g];
        if (!p) return fail(planners[0], LTP_ERR_INVALID_ARGUMENT, "null planner");
        hipError_t e = hipSetDevice(p->device);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)shards[g].stream);
        if (e != hipSuccess) {
            const int code = hip_fail(p, e, "hipStreamSynchronize");
            return g ? fail(planners[0], code, "shard " + std::to_string(g) + ": " + p->err) : code;
        }
    }
    return LTP_OK;
}

int ltp_plan_envelope_multi_host(ltp_planner* const* planners, int k, long long n, const double* q_goal, const double* q_0,
                                 const double* v_0, const double* a_0, int window, int n_windows, const ltp_records* host_records,
                                 double* env)
{
    if (!planners || k < 1 || !planners[0]) return LTP_ERR_INVALID_ARGUMENT;
    ltp_planner* p0 = planners[0];
    if (n < 0 || !env) return fail(p0, LTP_ERR_INVALID_ARGUMENT, "null argument");
    int rc = check_shard_planners(planners, k);
    if (rc != LTP_OK) return rc;
    const int dof = p0->dof;
    return run_shards(planners, k, [&](int g) -> int {
        long long f = 0, c = 0;
        ltp_shard_range(n, g, k, &f, &c);
        const size_t fd = (size_t)f * dof;
        ltp_records r{};
        if (host_records) {
            r = *host_records;
            if (r.t_opt) r.t_opt += fd * 7;
            if (r.t_scaled) r.t_scaled += fd * 7;
            if (r.dir) r.dir += fd;
            if (r.v_drive) r.v_drive += fd;
            if (r.mod) r.mod += fd;
            if (r.t_required) r.t_required += f;
            if (r.slowest) r.slowest += f;
            if (r.traj_len) r.traj_len += f;
            if (r.status) r.status += f;
        }
        return ltp_plan_envelope_host(planners[g], c, q_goal ? q_goal + fd : nullptr, q_0 ? q_0 + fd : nullptr, v_0 ? v_0 + fd : nullptr,
                                      a_0 ? a_0 + fd : nullptr, window, n_windows, host_records ? &r : nullptr,
                                      env + fd * (size_t)n_windows * 2);
    });
}

int ltp_plan_envelope_host(ltp_planner* p, long long n, const double* q_goal, const double* q_0, const double* v_0,
                           const double* a_0, int window, int n_windows, const ltp_records* host_records, double* env)
{
    if (!p || n < 0 || !env) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    if (window < 1 || n_windows < 1) return fail(p, LTP_ERR_INVALID_ARGUMENT, "window and n_windows must be >= 1");
    if (n > 0 && p->dof > 0 && (!q_goal || !q_0 || !v_0 || !a_0)) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null query array");
    int rc;
    { std::lock_guard<std::mutex> g(p->mu); rc = check_config(p); }
    if (rc != LTP_OK) return rc;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    const int dof = p->dof;
    const size_t nd = (size_t)n * dof;
    const double* const h_in[4] = {q_goal, q_0, v_0, a_0};
    DevRecords dr;
    LTP_HIP_TRY(p, dr.alloc_all(n, dof));
    double* d_in[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int k = 0; k < 4; ++k) {
        LTP_HIP_TRY(p, dr.alloc(&d_in[k], nd));
        if (nd) LTP_HIP_TRY(p, hipMemcpy(d_in[k], h_in[k], sizeof(double) * nd, hipMemcpyHostToDevice));
    }
    double* d_env = nullptr;
    const size_t env_doubles = nd * (size_t)n_windows * 2;
    LTP_HIP_TRY(p, dr.alloc(&d_env, env_doubles));
    const ltp_queries dq{d_in[0], d_in[1], d_in[2], d_in[3], dof, 1};
    rc = ltp_plan_switch_times_batch(p, n, &dq, &dr.r, nullptr, nullptr);
    if (rc == LTP_OK) rc = ltp_envelope_batch(p, 0, n, &dq, &dr.r, window, n_windows, d_env, nullptr);
    if (rc != LTP_OK) return rc;
    if (env_doubles) LTP_HIP_TRY(p, hipMemcpy(env, d_env, sizeof(double) * env_doubles, hipMemcpyDeviceToHost));   // synchronises
    else LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
    return download_records(p, n, dof, dr.r, host_records);   // after the consumer: status carries END_LIMIT
}

int ltp_get_trajectory_host(ltp_planner* p, long long n, const double* t, const double* dir, const signed char* mod,
                            const double* q_0, const double* v_0, const double* a_0, const double* v_drive,
                            int* traj_len, int* status, unsigned long long* offsets, double** packed)
{
    if (!p || n < 0 || !offsets || !packed || (n > 0 && (!t || !dir || !mod || !q_0 || !v_0 || !a_0 || !v_drive)))
        return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    *packed = nullptr;
    int rc;
    { std::lock_guard<std::mutex> g(p->mu); rc = check_config(p); if (rc == LTP_OK) rc = reserve(p, n > 0 ? n : 1); }
    if (rc != LTP_OK) return rc;
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    const int dof = p->dof;
    const size_t nd = (size_t)n * dof;
    if (n > 0 && dof > 0 && arena_layout(n, dof).end <= kSmallHostBytes) {
        // staged path: persistent arena + pinned mirror, one upload, one download
        std::lock_guard<std::mutex> hg(p->host_mu);
        if (nd <= (size_t)ltp::small_batch_pairs() && p->semantics == LTP_SEMANTICS_CPP) {
            // fused path: one launch, rows written straight into the pinned result buffer
            const double* const h_in[4] = {nullptr, q_0, v_0, a_0};
            const ltp_records given{nullptr, const_cast<double*>(t), const_cast<double*>(dir), const_cast<double*>(v_drive),
                                    const_cast<signed char*>(mod), nullptr, nullptr, nullptr, nullptr};
            const ltp_records outr{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, traj_len, status};
            bool handled = false;
            rc = plan_batch_host_fused(p, n, h_in, &outr, &given, offsets, packed, &handled);
            if (rc != LTP_OK || handled) return rc;
        }
        const ArenaLayout L = arena_layout(n, dof);
        rc = ensure_arena(p, L.end);
        if (rc != LTP_OK) return rc;
        memset(p->h_arena, 0, L.end);
        memcpy(p->h_arena + L.in[1], q_0, sizeof(double) * nd);
        memcpy(p->h_arena + L.in[2], v_0, sizeof(double) * nd);
        memcpy(p->h_arena + L.in[3], a_0, sizeof(double) * nd);
        memcpy(p->h_arena + L.t_scaled, t, sizeof(double) * nd * 7);
        memcpy(p->h_arena + L.dir, dir, sizeof(double) * nd);
        memcpy(p->h_arena + L.v_drive, v_drive, sizeof(double) * nd);
        memcpy(p->h_arena + L.mod, mod, nd);
        LTP_HIP_TRY(p, hipMemcpyAsync(p->d_arena, p->h_arena, L.end, hipMemcpyHostToDevice, nullptr));
        const ltp_queries dq{(double*)(p->d_arena + L.in[1]), (double*)(p->d_arena + L.in[1]), (double*)(p->d_arena + L.in[2]),
                             (double*)(p->d_arena + L.in[3]), dof, 1};   // q_goal is not used by the sampler
        const ltp_records dr = arena_records(p->d_arena, L);
        unsigned long long* d_off = (unsigned long long*)(p->d_arena + L.offsets);
        {
            std::lock_guard<std::mutex> g(p->mu);
            capture_geometry(p);
            ltp::launch_offsets(nullptr, n, dof, p->t_sample, to_dev(&dr), p->d_block_sums, d_off, false,
                                ltp::RowSpec{p->max_samples, p->sample_stride});
            LTP_HIP_TRY(p, hipGetLastError());
        }
        LTP_HIP_TRY(p, hipMemcpyAsync(p->h_arena + L.rec_begin, p->d_arena + L.rec_begin, L.end - L.rec_begin, hipMemcpyDeviceToHost, nullptr));
        LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
        const unsigned long long* h_off = (const unsigned long long*)(p->h_arena + L.offsets);
        rc = sample_to_host_small(p, n, L, dq, dr, d_off, h_off[n], packed);
        if (rc != LTP_OK) return rc;
        memcpy(offsets, h_off, sizeof(unsigned long long) * ((size_t)n + 1));
        if (traj_len) memcpy(traj_len, p->h_arena + L.traj_len, sizeof(int) * (size_t)n);
        if (status) memcpy(status, p->h_arena + L.status, sizeof(int) * (size_t)n);
        return LTP_OK;
    }
    DevRecords dr;
    LTP_HIP_TRY(p, dr.alloc_all(n, dof));
    if (nd) {
        LTP_HIP_TRY(p, hipMemcpy(dr.r.t_scaled, t, sizeof(double) * nd * 7, hipMemcpyHostToDevice));
        LTP_HIP_TRY(p, hipMemcpy(dr.r.dir, dir, sizeof(double) * nd, hipMemcpyHostToDevice));
        LTP_HIP_TRY(p, hipMemcpy(dr.r.mod, mod, nd, hipMemcpyHostToDevice));
        LTP_HIP_TRY(p, hipMemcpy(dr.r.v_drive, v_drive, sizeof(double) * nd, hipMemcpyHostToDevice));
    }
    if (n) LTP_HIP_TRY(p, hipMemset(dr.r.status, 0, sizeof(int) * (size_t)n));
    double* d_in[3] = {nullptr, nullptr, nullptr};
    const double* h_in[3] = {q_0, v_0, a_0};
    for (int k = 0; k < 3; ++k) {
        LTP_HIP_TRY(p, dr.alloc(&d_in[k], nd));
        if (nd) LTP_HIP_TRY(p, hipMemcpy(d_in[k], h_in[k], sizeof(double) * nd, hipMemcpyHostToDevice));
    }
    unsigned long long* d_off = nullptr;
    LTP_HIP_TRY(p, dr.alloc(&d_off, (size_t)n + 1));
    LTP_HIP_TRY(p, hipMemset(d_off, 0, sizeof(unsigned long long) * ((size_t)n + 1)));
    ltp_queries dq{d_in[0], d_in[0], d_in[1], d_in[2], dof, 1};   // q_goal is not used by the sampler
    if (n > 0 && dof > 0) {
        std::lock_guard<std::mutex> g(p->mu);
        capture_geometry(p);
        ltp::launch_offsets(nullptr, n, dof, p->t_sample, to_dev(&dr.r), p->d_block_sums, d_off, false, ltp::RowSpec{p->max_samples, p->sample_stride});
        LTP_HIP_TRY(p, hipGetLastError());
    }
    LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
    rc = run_sample_to_host(p, n, dq, dr.r, d_off, offsets, packed);
    if (rc != LTP_OK) return rc;
    if (traj_len) LTP_HIP_TRY(p, hipMemcpy(traj_len, dr.r.traj_len, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    if (status) LTP_HIP_TRY(p, hipMemcpy(status, dr.r.status, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    return LTP_OK;
}

void ltp_free_host(void* ptr)
{
    if (ptr && !g_pinned.release(ptr)) free(ptr);
}

int ltp_check_inputs_host(ltp_planner* p, const double* q_0, const double* v_0, const double* a_0, int* ok)
{
    if (!p || !ok) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    int rc;
    { std::lock_guard<std::mutex> g(p->mu); rc = check_config(p); }
    if (rc != LTP_OK) return rc;
    const int dof = p->dof;
    if (dof == 0) { *ok = 1; return LTP_OK; }   // the reference's loop over zero joints (cc:72-76)
    if (!q_0 || !v_0 || !a_0) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    std::lock_guard<std::mutex> hg(p->host_mu);
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    const size_t row = ((sizeof(double) * dof) + 15) & ~(size_t)15;
    rc = ensure_arena(p, 3 * row + 16);
    if (rc != LTP_OK) return rc;
    memcpy(p->h_arena, q_0, sizeof(double) * dof);
    memcpy(p->h_arena + row, v_0, sizeof(double) * dof);
    memcpy(p->h_arena + 2 * row, a_0, sizeof(double) * dof);
    ltp::launch_check_inputs(nullptr, dof, dev_limits(p), (const double*)p->h_arena, (const double*)(p->h_arena + row),
                             (const double*)(p->h_arena + 2 * row), (int*)(p->h_arena + 3 * row), p->semantics);   // pinned: no copies
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
    *ok = *(const int*)(p->h_arena + 3 * row);
    return LTP_OK;
}

int ltp_opt_braking_host(ltp_planner* p, int joint, double v_0, double a_0, double* q, double* t_rel, double* dir)
{
    if (!p || !q || !t_rel || !dir) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    double buf[16] = {0};
    memcpy(buf, t_rel, sizeof(double) * 7);
    const int rc = run_one_lane(p, joint, buf, [&](double* io, double ts, const ltp::Limits& lim, int sem) { ltp::launch_single_opt_braking(nullptr, joint, ts, lim, v_0, a_0, io, sem); });
    if (rc != LTP_OK) return rc;
    memcpy(t_rel, buf, sizeof(double) * 7);
    *q = buf[7];
    *dir = buf[8];
    return LTP_OK;
}

int ltp_opt_switch_times_host(ltp_planner* p, int joint, double q_goal, double q_0, double v_0, double a_0, double v_drive,
                              double* t, double* dir, char* mod, int* ok)
{
    if (!p || !t || !dir || !mod || !ok) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    double buf[16] = {0};
    memcpy(buf, t, sizeof(double) * 7);
    const int rc = run_one_lane(p, joint, buf, [&](double* io, double ts, const ltp::Limits& lim, int sem) { ltp::launch_single_opt_switch(nullptr, joint, ts, lim, q_goal, q_0, v_0, a_0, v_drive, io, sem); });
    if (rc != LTP_OK) return rc;
    memcpy(t, buf, sizeof(double) * 7);
    *dir = buf[7];
    *mod = (char)(int)buf[8];
    *ok = (int)buf[9];
    return LTP_OK;
}

int ltp_time_scaling_host(ltp_planner* p, int joint, double q_goal, double q_0, double v_0, double a_0, double dir,
                          double t_required, double* scaled_t, double* v_drive, char* mod, int* ok, int* accepted_case)
{
    if (!p || !scaled_t || !v_drive || !mod || !ok) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    double buf[16] = {0};
    memcpy(buf, scaled_t, sizeof(double) * 7);
    const int rc = run_one_lane(p, joint, buf, [&](double* io, double ts, const ltp::Limits& lim, int sem) { ltp::launch_single_time_scaling(nullptr, joint, ts, lim, q_goal, q_0, v_0, a_0, dir, t_required, io, sem); });
    if (rc != LTP_OK) return rc;
    memcpy(scaled_t, buf, sizeof(double) * 7);
    *v_drive = buf[7];
    *mod = (char)(int)buf[8];
    *ok = (int)buf[9];
    if (accepted_case) *accepted_case = (int)buf[10];
    return LTP_OK;
}

static int roots_host_any(ltp_planner* p, long long n, int degree, bool f32, const void* coef, void* re, void* im)
{
    if (!p || n < 0 || degree < 1 || degree > 8 || !coef || !re || !im) return fail(p, LTP_ERR_INVALID_ARGUMENT, "bad argument (degree 1..8)");
    // long_term_planner/roots.h routes every roots() call of a process through one handle: serialise them, and stage through
    // the handle's pinned arena (host memory the kernel reads and writes directly) instead of three allocations and copies per call
    std::lock_guard<std::mutex> hg(p->host_mu);
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    if (n == 0) return LTP_OK;
    const size_t es = f32 ? sizeof(float) : sizeof(double);
    const size_t cb = ((size_t)n * (degree + 1) * es + 15) & ~(size_t)15, rb = ((size_t)n * degree * es + 15) & ~(size_t)15;
    if (cb + 2 * rb <= kSmallHostBytes) {
        int rc = ensure_arena(p, cb + 2 * rb);
        if (rc != LTP_OK) return rc;
        memcpy(p->h_arena, coef, (size_t)n * (degree + 1) * es);
        ltp::launch_roots_all(nullptr, n, degree, f32, p->h_arena, p->h_arena + cb, p->h_arena + cb + rb);
        LTP_HIP_TRY(p, hipGetLastError());
        LTP_HIP_TRY(p, hipStreamSynchronize(nullptr));
        memcpy(re, p->h_arena + cb, (size_t)n * degree * es);
        memcpy(im, p->h_arena + cb + rb, (size_t)n * degree * es);
        return LTP_OK;
    }
    void *dc = nullptr, *dr = nullptr, *di = nullptr;
    DevRecords holder;
    LTP_HIP_TRY(p, holder.alloc((char**)&dc, (size_t)n * (degree + 1) * es));
    LTP_HIP_TRY(p, holder.alloc((char**)&dr, (size_t)n * degree * es));
    LTP_HIP_TRY(p, holder.alloc((char**)&di, (size_t)n * degree * es));
    LTP_HIP_TRY(p, hipMemcpy(dc, coef, (size_t)n * (degree + 1) * es, hipMemcpyHostToDevice));
    ltp::launch_roots_all(nullptr, n, degree, f32, dc, dr, di);
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipMemcpy(re, dr, (size_t)n * degree * es, hipMemcpyDeviceToHost));
    LTP_HIP_TRY(p, hipMemcpy(im, di, (size_t)n * degree * es, hipMemcpyDeviceToHost));
    return LTP_OK;
}

int ltp_roots_f64_host(ltp_planner* p, long long n, int degree, const double* coef, double* re, double* im)
{
    return roots_host_any(p, n, degree, false, coef, re, im);
}

int ltp_roots_f32_host(ltp_planner* p, long long n, int degree, const float* coef, float* re, float* im)
{
    return roots_host_any(p, n, degree, true, coef, re, im);
}

int ltp_debug_last_matlab_flags(const ltp_planner* p) { return p ? p->last_matlab_flags : -1; }

int ltp_debug_roots_matlab_host(ltp_planner* p, long long n, int degree, const double* coef, double* re, double* im, int* nroots, int* status)
{
    if (!p || n < 0 || degree < 1 || degree > 6 || !coef || !re || !im || !nroots || !status) return fail(p, LTP_ERR_INVALID_ARGUMENT, "bad argument (degree 1..6)");
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    double *dc = nullptr, *dr = nullptr, *di = nullptr;
    int *dn = nullptr, *ds = nullptr;
    DevRecords holder;
    LTP_HIP_TRY(p, holder.alloc(&dc, (size_t)n * (degree + 1)));
    LTP_HIP_TRY(p, holder.alloc(&dr, (size_t)n * degree));
    LTP_HIP_TRY(p, holder.alloc(&di, (size_t)n * degree));
    LTP_HIP_TRY(p, holder.alloc(&dn, (size_t)n));
    LTP_HIP_TRY(p, holder.alloc(&ds, (size_t)n));
    LTP_HIP_TRY(p, hipMemcpy(dc, coef, sizeof(double) * (size_t)n * (degree + 1), hipMemcpyHostToDevice));
    ltp::launch_roots_matlab(nullptr, n, degree, dc, dr, di, dn, ds);
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipMemcpy(re, dr, sizeof(double) * (size_t)n * degree, hipMemcpyDeviceToHost));
    LTP_HIP_TRY(p, hipMemcpy(im, di, sizeof(double) * (size_t)n * degree, hipMemcpyDeviceToHost));
    LTP_HIP_TRY(p, hipMemcpy(nroots, dn, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
    LTP_HIP_TRY(p, hipMemcpy(status, ds, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost));
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

int ltp_debug_math_probe_host(ltp_planner* p, long long n, const double* x, const double* y, double* out)
{
    if (!p || n < 0 || !x || !y || !out) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    DevRecords holder;
    LTP_HIP_TRY(p, holder.alloc(&dx, (size_t)n));
    LTP_HIP_TRY(p, holder.alloc(&dy, (size_t)n));
    LTP_HIP_TRY(p, holder.alloc(&dout, (size_t)n * 8));
    LTP_HIP_TRY(p, hipMemcpy(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    LTP_HIP_TRY(p, hipMemcpy(dy, y, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    ltp::launch_math_probe(nullptr, n, dx, dy, dout);
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipMemcpy(out, dout, sizeof(double) * (size_t)n * 8, hipMemcpyDeviceToHost));
    return LTP_OK;
}

int ltp_debug_roots_probe_host(ltp_planner* p, long long n, int degree, const double* coef, double* root)
{
    if (!p || n < 0 || !coef || !root || degree < 4 || degree > 6) return fail(p, LTP_ERR_INVALID_ARGUMENT, "bad argument");
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    double *dc = nullptr, *dr = nullptr;
    DevRecords holder;
    LTP_HIP_TRY(p, holder.alloc(&dc, (size_t)n * 7));
    LTP_HIP_TRY(p, holder.alloc(&dr, (size_t)n));
    LTP_HIP_TRY(p, hipMemcpy(dc, coef, sizeof(double) * (size_t)n * 7, hipMemcpyHostToDevice));
    ltp::launch_roots_probe(nullptr, n, degree, dc, dr);
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipMemcpy(root, dr, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    return LTP_OK;
}

}  // extern "C"
