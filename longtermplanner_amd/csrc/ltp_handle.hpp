// ltp_handle.hpp — the planner handle behind include/ltp_hip.h and the helpers the C-ABI translation units share
// (ltp_capi_handle.hip: lifetime / configuration / workspace; ltp_capi_batch.hip: device-pointer batch calls;
// ltp_capi_host.hip: host-pointer convenience calls; ltp_capi_multi.hip: one process, several shards). Host-side only; every
// computation is a kernel in the ltp_*.hip kernel files. There is deliberately no CPU implementation behind the entry points:
// without a HIP device they fail with LTP_ERR_NO_DEVICE.
//
// Locks: host_mu (the synchronous host-pointer calls and their arena) is taken BEFORE mu (configuration and the device
// workspace), everywhere.
#pragma once
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
    int walk_blocks[2] = {0, 0};              // resident blocks of k_sample_walk f64 / f32
    int walk_auto_cus = 0;                    // compute units of `device`, set with the autonomous-wave kernels' LDS limit (reserve())
    unsigned long long tables_cap = 4ull << 30;   // upper bound for d_tables (ltp_create: 1/16 of the device's memory if that
                                                  // is more — 18 GiB of 288); longer ranges are processed in pieces
    double* d_small = nullptr;             // 16 doubles for the one-lane entry points
    bool small_dirty = false;              // a fused small-batch call failed: k_plan_small's arrival word may be non-zero
    const char* last_kernel = "";          // row / envelope kernel of the latest ltp_sample_batch* / ltp_envelope_batch
    int semantics = 0;                     // LTP_SEMANTICS_CPP (the reference's C++, default) or LTP_SEMANTICS_MATLAB
    int envelope_mode = LTP_ENVELOPE_ANALYTIC;   // the default since round 6 (8.8e9 window values identical to the exhaustive form, profiles/r06_envelope_mode_soak.json); LTP_ENVELOPE_EXHAUSTIVE: bit-identical to the reduced rows by construction
    int pow_rule = LTP_POW_LIBM;           // LTP_POW_LIBM (default) or LTP_POW_EXACT: how pow(x, 3 | 4 | 6 | 1/2) is formed (ltp_math.hpp)
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

namespace ltp_capi {

constexpr bool kEnvelopeTablePassByDefault = true;   // measured: see DESIGN.md "Table pass"

int fail(ltp_planner* p, int code, const std::string& msg);
int hip_fail(ltp_planner* p, hipError_t e, const char* what);
#define LTP_HIP_TRY(p, expr)                                              \
    do {                                                                  \
        hipError_t e_ = (expr);                                           \
        if (e_ != hipSuccess) return ltp_capi::hip_fail((p), e_, #expr);  \
    } while (0)

int upload_limits(ltp_planner* p);
ltp::Limits dev_limits(const ltp_planner* p);
// the template variant of the stage kernels: semantics (bit 0) | pow rule (bit 1), see ltp::dispatch_variant
inline int stage_variant(const ltp_planner* p) { return p->semantics | (p->pow_rule == LTP_POW_LIBM ? 2 : 0); }
int check_config(ltp_planner* p);                        // the reference indexes its limit vectors unchecked (UB when short); here it is an error
int reserve(ltp_planner* p, long long n);
ltp::Queries to_dev(const ltp_queries* in);
ltp::Records to_dev(const ltp_records* r);
bool records_complete(const ltp_records* r);
int workspace_acquire(ltp_planner* p, hipStream_t s, bool& capturing);
int workspace_release(ltp_planner* p, hipStream_t s, bool capturing);
void capture_geometry(ltp_planner* p);
int check_geometry(ltp_planner* p);
bool want_table_pass(const ltp_planner* p, unsigned long long row_bytes, bool f32);
bool want_walk(const ltp_planner* p, int max_samples, int stride, bool f32);
int ensure_tables(ltp_planner* p, long long count, bool capturing, long long* plans_per_piece);

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

}  // namespace ltp_capi
