// ltp_capi_multi.hip — C ABI (include/ltp_hip.h): one process, several shards: contiguous query ranges, one host
// thread per shard, no collective.
#include "ltp_handle.hpp"

using namespace ltp_capi;

extern "C" {

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
                    pg->sample_stride == p0->sample_stride && pg->goal_check == p0->goal_check && pg->semantics == p0->semantics &&
                    pg->pow_rule == p0->pow_rule && pg->envelope_mode == p0->envelope_mode;
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

}  // extern "C"
