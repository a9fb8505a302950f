// ltp_capi_host.hip — C ABI (include/ltp_hip.h): host-pointer convenience calls (synchronous): whole batches staged through
// device memory, the one-launch single call (k_plan_small), the reference's protected methods as one-lane calls, roots().
#include "ltp_handle.hpp"

using namespace ltp_capi;

extern "C" {

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
                               (int*)(p->h_arena + ends_at), (unsigned int*)p->d_small, done, given != nullptr, p->pow_rule == LTP_POW_LIBM, true);
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
        semantics = stage_variant(p);                  // semantics | pow rule << 1
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
                             (const double*)(p->h_arena + 2 * row), (int*)(p->h_arena + 3 * row), p->semantics);   // pinned: no copies (checkInputs forms no powers)
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

int ltp_debug_libm_pow_host(ltp_planner* p, long long n, const double* x, const double* y, double* out)
{
    if (!p || n < 0 || !x || !y || !out) return fail(p, LTP_ERR_INVALID_ARGUMENT, "null argument");
    LTP_HIP_TRY(p, hipSetDevice(p->device));
    double *dx = nullptr, *dy = nullptr, *dout = nullptr;
    DevRecords holder;
    LTP_HIP_TRY(p, holder.alloc(&dx, (size_t)n));
    LTP_HIP_TRY(p, holder.alloc(&dy, (size_t)n));
    LTP_HIP_TRY(p, holder.alloc(&dout, (size_t)n));
    LTP_HIP_TRY(p, hipMemcpy(dx, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    LTP_HIP_TRY(p, hipMemcpy(dy, y, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
    ltp::launch_libm_pow_probe(nullptr, n, dx, dy, dout);
    LTP_HIP_TRY(p, hipGetLastError());
    LTP_HIP_TRY(p, hipMemcpy(out, dout, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
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
