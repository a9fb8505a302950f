// device_api_test — the C ABI of include/ltp_hip.h driven from plain C++ with hipMalloc'd buffers (no PyTorch in the
// process), the way a C++ user of the reference would call the batched path:
//   1. generate / plan / sample / reduce a batch through the device-pointer entry points;
//   2. the host-pointer convenience path (ltp_plan_batch_host) must return the same bits for the same queries;
//   3. the plan + sample sequence is captured into a hipGraph after ltp_reserve_batch and replayed on new inputs in
//      place; the replay must equal the eager calls bit for bit;
//   4. the envelope consumer equals min / max over windows of the sampled rows;
//   6. the table-pass calls (envelopes, capped rows) inside a capture: refused on a handle without a table workspace (nothing
//      may be allocated while capturing), captured and replayed in pieces after ltp_reserve_tables;
//   7. several host threads on ONE handle (plans, one-joint calls, checkInputs, ltp_set_limits): own results, no deadlock.
// Built with g++ against the HIP runtime API only; prints "0 failures" on success.
#include <hip/hip_runtime_api.h>

#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <thread>
#include <vector>

#include "ltp_hip.h"

static int g_failures = 0;
#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            ++g_failures;                                                        \
            std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond);        \
        }                                                                        \
    } while (0)
#define HIP(call)                                                                \
    do {                                                                         \
        hipError_t e_ = (call);                                                  \
        if (e_ != hipSuccess) {                                                  \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                            \
        }                                                                        \
    } while (0)
#define LTP(call)                                                                \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != LTP_OK) {                                                     \
            std::printf("ltp error %d (%s) at %s:%d\n", rc_, ltp_last_error(h), __FILE__, __LINE__); \
            return 2;                                                            \
        }                                                                        \
    } while (0)

template <class T>
static std::vector<T> download(const T* d, size_t n)
{
    std::vector<T> v(n);
    if (n && hipMemcpy(v.data(), d, n * sizeof(T), hipMemcpyDeviceToHost) != hipSuccess) std::printf("download failed\n"), ++g_failures;
    return v;
}

int main()
{
    const int dof = 7;
    const long long n = 4000;
    // the panda-like limit set of SURVEY.md §8(d)
    const double q_min[dof] = {-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973};
    const double q_max[dof] = {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973};
    const double v_max[dof] = {2.175, 2.175, 2.175, 2.175, 2.61, 2.61, 2.61};
    const double a_max[dof] = {15, 7.5, 10, 12.5, 15, 20, 20};
    const double j_max[dof] = {7500, 3750, 5000, 6250, 7500, 10000, 10000};
    ltp_planner* h = nullptr;
    int rc = ltp_create(dof, 0.001, q_min, q_max, v_max, a_max, j_max, 0, &h);
    if (rc != LTP_OK) {
        std::printf("ltp_create failed with %d (no HIP device? the library has no CPU fallback)\n", rc);
        return 2;
    }
    hipStream_t s;
    HIP(hipStreamCreate(&s));

    const size_t nd = (size_t)n * dof;
    double *in[4], *t_opt, *t_scaled, *dir, *v_drive, *t_required;
    signed char* mod;
    int *slowest, *traj_len, *status;
    unsigned long long* offsets;
    for (auto& p : in) HIP(hipMalloc((void**)&p, nd * sizeof(double)));
    HIP(hipMalloc((void**)&t_opt, nd * 7 * sizeof(double)));
    HIP(hipMalloc((void**)&t_scaled, nd * 7 * sizeof(double)));
    HIP(hipMalloc((void**)&dir, nd * sizeof(double)));
    HIP(hipMalloc((void**)&v_drive, nd * sizeof(double)));
    HIP(hipMalloc((void**)&mod, nd));
    HIP(hipMalloc((void**)&t_required, n * sizeof(double)));
    HIP(hipMalloc((void**)&slowest, n * sizeof(int)));
    HIP(hipMalloc((void**)&traj_len, n * sizeof(int)));
    HIP(hipMalloc((void**)&status, n * sizeof(int)));
    HIP(hipMalloc((void**)&offsets, (n + 1) * sizeof(unsigned long long)));
    const ltp_queries q{in[0], in[1], in[2], in[3], dof, 1};
    const ltp_records rec{t_opt, t_scaled, dir, v_drive, mod, t_required, slowest, traj_len, status};

    // ---- 1. device path ----
    LTP(ltp_reserve_batch(h, n));
    LTP(ltp_generate_queries_batch(h, n, 1, 0, in[0], in[1], in[2], in[3], dof, 1, s));
    LTP(ltp_plan_switch_times_batch(h, n, &q, &rec, offsets, s));
    HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> off = download(offsets, (size_t)n + 1);
    // room for any batch of this size: rows are padded to 32 samples, plans last a few seconds at most
    const unsigned long long capacity = off[n] + off[n] / 2;
    double *tile, *tile2;
    HIP(hipMalloc((void**)&tile, capacity * sizeof(double)));
    HIP(hipMalloc((void**)&tile2, capacity * sizeof(double)));
    LTP(ltp_sample_batch(h, 0, n, &q, &rec, offsets, tile, capacity, 1, s));
    const int W = 50, K = 40;
    double* env;
    HIP(hipMalloc((void**)&env, nd * K * 2 * sizeof(double)));
    LTP(ltp_envelope_batch(h, 0, n, &q, &rec, W, K, env, s));
    HIP(hipStreamSynchronize(s));
    std::vector<double> rows = download(tile, (size_t)off[n]);
    std::vector<int> len = download(traj_len, (size_t)n), st = download(status, (size_t)n);
    std::vector<double> ts = download(t_scaled, nd * 7);
    long long ok = 0;
    for (long long p = 0; p < n; ++p) ok += st[p] == 0;
    std::printf("device path: %lld plans, %lld ok, %.1f MB of trajectories\n", n, ok, off[n] * 8 / 1e6);
    CHECK(ok > n * 9 / 10);

    // ---- 2. host-pointer path on the same queries ----
    std::vector<double> hq[4];
    for (int k = 0; k < 4; ++k) hq[k] = download(in[k], nd);
    {
        std::vector<double> h_topt(nd * 7), h_ts(nd * 7), h_dir(nd), h_vd(nd), h_treq(n);
        std::vector<signed char> h_mod(nd);
        std::vector<int> h_slow(n), h_len(n), h_st(n);
        std::vector<unsigned long long> h_off(n + 1);
        ltp_records hr{h_topt.data(), h_ts.data(), h_dir.data(), h_vd.data(), h_mod.data(), h_treq.data(), h_slow.data(), h_len.data(), h_st.data()};
        double* packed = nullptr;
        LTP(ltp_plan_batch_host(h, n, hq[0].data(), hq[1].data(), hq[2].data(), hq[3].data(), &hr, h_off.data(), &packed));
        CHECK(h_off == off);
        CHECK(h_len == len);
        CHECK(h_st == st);
        CHECK(std::memcmp(h_ts.data(), ts.data(), nd * 7 * sizeof(double)) == 0);
        // compare the rows, not the padding between them (the host path returns a zero-initialised buffer)
        size_t bad = 0;
        for (long long p = 0; p < n; ++p) {
            const int L = len[p];
            if (L <= 0) continue;
            const size_t stride = (size_t)ltp_row_stride(L);
            for (int r = 0; r < 4 * dof; ++r)
                bad += std::memcmp(packed + off[p] + r * stride, rows.data() + off[p] + r * stride, (size_t)L * sizeof(double)) != 0;
        }
        CHECK(bad == 0);
        ltp_free_host(packed);
    }

    // ---- 4. envelope == min / max over windows of the sampled q rows ----
    {
        std::vector<double> e = download(env, nd * K * 2);
        size_t bad = 0;
        for (long long p = 0; p < n; p += 37) {
            const int L = len[p];
            for (int j = 0; j < dof; ++j)
                for (int w = 0; w < K; ++w) {
                    const double lo = e[((p * dof + j) * K + w) * 2], hi = e[((p * dof + j) * K + w) * 2 + 1];
                    if (L <= 0) {
                        bad += !(std::isnan(lo) && std::isnan(hi));
                        continue;
                    }
                    const double* row = rows.data() + off[p] + (size_t)j * ltp_row_stride(L);   // q block comes first
                    const int b = std::min(w * W, L - 1), en = std::min(w * W + W, L);
                    double mn = row[b], mx = row[b];
                    for (int i = b; i < std::max(en, b + 1); ++i) { mn = std::min(mn, row[i]); mx = std::max(mx, row[i]); }
                    bad += !(mn == lo && mx == hi);
                }
        }
        CHECK(bad == 0);
    }

    // ---- 3. hipGraph: capture plan + sample, replay on new inputs in place ----
    {
        hipGraph_t graph;
        hipGraphExec_t exec;
        HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        int rc1 = ltp_plan_switch_times_batch(h, n, &q, &rec, offsets, s);
        int rc2 = ltp_sample_batch(h, 0, n, &q, &rec, offsets, tile, capacity, 1, s);
        HIP(hipStreamEndCapture(s, &graph));
        CHECK(rc1 == LTP_OK && rc2 == LTP_OK);
        HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (unsigned long long seed = 2; seed <= 4; ++seed) {
            LTP(ltp_generate_queries_batch(h, n, seed, 0, in[0], in[1], in[2], in[3], dof, 1, s));
            HIP(hipMemsetAsync(tile, 0, capacity * sizeof(double), s));
            HIP(hipGraphLaunch(exec, s));
            HIP(hipStreamSynchronize(s));
            std::vector<unsigned long long> off_g = download(offsets, (size_t)n + 1);
            std::vector<int> st_g = download(status, (size_t)n);
            CHECK(off_g[n] <= capacity);
            std::vector<double> rows_g = download(tile, (size_t)off_g[n]);
            // the same, eagerly
            HIP(hipMemsetAsync(tile2, 0, capacity * sizeof(double), s));
            LTP(ltp_plan_switch_times_batch(h, n, &q, &rec, offsets, s));
            LTP(ltp_sample_batch(h, 0, n, &q, &rec, offsets, tile2, capacity, 1, s));
            HIP(hipStreamSynchronize(s));
            CHECK(download(offsets, (size_t)n + 1) == off_g);
            CHECK(download(status, (size_t)n) == st_g);
            std::vector<double> rows_e = download(tile2, (size_t)off_g[n]);
            CHECK(std::memcmp(rows_e.data(), rows_g.data(), rows_e.size() * sizeof(double)) == 0);
            CHECK(off_g != off);   // really new inputs
        }
        // replay rate: launch-bound host work is gone
        hipEvent_t e0, e1;
        HIP(hipEventCreate(&e0));
        HIP(hipEventCreate(&e1));
        HIP(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i) HIP(hipGraphLaunch(exec, s));
        HIP(hipEventRecord(e1, s));
        HIP(hipStreamSynchronize(s));
        float ms = 0;
        HIP(hipEventElapsedTime(&ms, e0, e1));
        std::printf("hipGraph replay of plan + sample, %lld plans: %.3f ms per replay\n", n, ms / 20);
        HIP(hipGraphExecDestroy(exec));
        HIP(hipGraphDestroy(graph));
    }

    // ---- 6. calls under capture that need (or do not need) the table workspace: envelopes take the table pass by default; rows
    //         capped at <= 256 samples take k_sample_walk_* (tables kept in the compute unit: no workspace at all) unless flag bit 2
    //         asks for the table-pass kernels ----
    {
        ltp_planner* h1 = h;
        ltp_planner* h = nullptr;                         // the LTP macro reports this handle's error text
        if (ltp_create(dof, 0.001, q_min, q_max, v_max, a_max, j_max, 0, &h) != LTP_OK) { std::printf("second ltp_create failed\n"); return 2; }
        const int cap_samples = 128;
        LTP(ltp_set_max_samples(h, cap_samples));
        LTP(ltp_reserve_batch(h, n));
        unsigned long long* offsets2;
        double *tile3, *tile4, *tile5, *env2, *env3;
        const unsigned long long cap3 = (unsigned long long)n * 4 * dof * (unsigned long long)ltp_row_stride(cap_samples);
        HIP(hipMalloc((void**)&offsets2, (n + 1) * sizeof(unsigned long long)));
        HIP(hipMalloc((void**)&tile3, cap3 * sizeof(double)));
        HIP(hipMalloc((void**)&tile4, cap3 * sizeof(double)));
        HIP(hipMalloc((void**)&tile5, cap3 * sizeof(double)));
        HIP(hipMalloc((void**)&env2, nd * K * 2 * sizeof(double)));
        HIP(hipMalloc((void**)&env3, nd * K * 2 * sizeof(double)));
        hipGraph_t graph;
        // (a) cold handle: the table workspace does not exist yet and must not be allocated inside the capture
        HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        const int ra = ltp_plan_switch_times_batch(h, n, &q, &rec, offsets2, s);
        // the default envelope form (analytic, k_envelope_walk: round 6) needs no workspace; the exhaustive form takes the table pass
        const int rb0 = ltp_envelope_batch(h, 0, n, &q, &rec, W, K, env2, s);
        CHECK(ltp_get_envelope_mode(h) == LTP_ENVELOPE_ANALYTIC && rb0 == LTP_OK);
        LTP(ltp_set_envelope_mode(h, LTP_ENVELOPE_EXHAUSTIVE));
        const int rb = ltp_envelope_batch(h, 0, n, &q, &rec, W, K, env2, s);
        const int rc4 = ltp_sample_batch(h, 0, n, &q, &rec, offsets2, tile3, cap3, 1 | 4, s);      // the table-pass kernels: need the workspace
        const int rc5 = ltp_sample_batch(h, 0, n, &q, &rec, offsets2, tile5, cap3, 1, s);          // default for these rows: no workspace needed
        HIP(hipStreamEndCapture(s, &graph));
        HIP(hipGraphDestroy(graph));
        CHECK(ra == LTP_OK && rb == LTP_ERR_INVALID_ARGUMENT && rc4 == LTP_ERR_INVALID_ARGUMENT && rc5 == LTP_OK);
        CHECK(std::strncmp(ltp_last_sampler_kernel(h), "k_sample_walk_f64", 17) == 0);
        // (b) a workspace bounded to 16 MiB, about a third of the batch: the captured calls run in pieces, nothing is allocated
        // or freed — neither inside the capture nor by the eager calls below (growing the workspace would free the buffer the
        // instantiated graph points to)
        LTP(ltp_set_table_workspace(h, 16ull << 20));
        LTP(ltp_reserve_tables(h, n));
        hipGraphExec_t exec;
        HIP(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        const int r1 = ltp_plan_switch_times_batch(h, n, &q, &rec, offsets2, s);
        const int r2 = ltp_envelope_batch(h, 0, n, &q, &rec, W, K, env2, s);
        const int r3 = ltp_sample_batch(h, 0, n, &q, &rec, offsets2, tile3, cap3, 1 | 4, s);
        CHECK(std::strncmp(ltp_last_sampler_kernel(h), "k_sample_tab", 12) == 0);
        const int r4 = ltp_sample_batch(h, 0, n, &q, &rec, offsets2, tile5, cap3, 1, s);
        HIP(hipStreamEndCapture(s, &graph));
        CHECK(r1 == LTP_OK && r2 == LTP_OK && r3 == LTP_OK && r4 == LTP_OK);
        CHECK(std::strncmp(ltp_last_sampler_kernel(h), "k_sample_walk_f64", 17) == 0);
        HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (unsigned long long seed = 7; seed <= 8; ++seed) {
            LTP(ltp_generate_queries_batch(h, n, seed, 0, in[0], in[1], in[2], in[3], dof, 1, s));
            HIP(hipMemsetAsync(tile3, 0, cap3 * sizeof(double), s));
            HIP(hipMemsetAsync(tile4, 0, cap3 * sizeof(double), s));
            HIP(hipMemsetAsync(tile5, 0, cap3 * sizeof(double), s));
            HIP(hipGraphLaunch(exec, s));
            HIP(hipStreamSynchronize(s));
            const std::vector<unsigned long long> off_g = download(offsets2, (size_t)n + 1);
            const std::vector<int> st_g = download(status, (size_t)n);
            CHECK(off_g[n] <= cap3 && off_g[n] > 0);
            const std::vector<double> rows_g = download(tile3, (size_t)off_g[n]), rows_w = download(tile5, (size_t)off_g[n]), env_g = download(env2, nd * K * 2);
            // the same eagerly, rows by the fused kernel for good measure
            LTP(ltp_plan_switch_times_batch(h, n, &q, &rec, offsets2, s));
            LTP(ltp_envelope_batch(h, 0, n, &q, &rec, W, K, env3, s));
            LTP(ltp_sample_batch(h, 0, n, &q, &rec, offsets2, tile4, cap3, 1 | 8, s));
            HIP(hipStreamSynchronize(s));
            CHECK(std::strcmp(ltp_last_sampler_kernel(h), "k_sample") == 0);
            CHECK(download(offsets2, (size_t)n + 1) == off_g);
            CHECK(download(status, (size_t)n) == st_g);
            const std::vector<double> rows_e = download(tile4, (size_t)off_g[n]), env_e = download(env3, nd * K * 2);
            CHECK(std::memcmp(rows_e.data(), rows_g.data(), rows_e.size() * sizeof(double)) == 0);
            CHECK(std::memcmp(rows_e.data(), rows_w.data(), rows_e.size() * sizeof(double)) == 0);
            CHECK(std::memcmp(env_e.data(), env_g.data(), env_e.size() * sizeof(double)) == 0);
        }
        HIP(hipGraphExecDestroy(exec));
        HIP(hipGraphDestroy(graph));
        ltp_destroy(h);
        (void)h1;
    }

    // ---- 5. latency of one planTrajectory-sized call through the host-pointer path (config 1 of BASELINE.json) ----
    {
        std::vector<double> t_o(dof * 7), t_s(dof * 7), d(dof), vd(dof);
        std::vector<signed char> md(dof);
        double treq;
        int slow, ln, stt;
        unsigned long long o2[2];
        ltp_records hr{t_o.data(), t_s.data(), d.data(), vd.data(), md.data(), &treq, &slow, &ln, &stt};
        for (int variant = 0; variant < 4; ++variant) {
            // both pow rules (ltp_set_pow_rule): the default restates glibc's pow (~25 of them per joint through two table reads each)
            const int with_rows = variant & 1;
            CHECK(ltp_set_pow_rule(h, variant < 2 ? LTP_POW_LIBM : LTP_POW_EXACT) == LTP_OK);
            float best = 1e30f, sum = 0;
            const int reps = 200;
            for (int i = 0; i < reps + 20; ++i) {
                const long long pq = (i * 37) % n;
                double* packed = nullptr;
                timespec t0, t1;
                clock_gettime(CLOCK_MONOTONIC, &t0);
                int rc3 = ltp_plan_batch_host(h, 1, hq[0].data() + pq * dof, hq[1].data() + pq * dof, hq[2].data() + pq * dof,
                                              hq[3].data() + pq * dof, &hr, o2, with_rows ? &packed : nullptr);
                clock_gettime(CLOCK_MONOTONIC, &t1);
                if (packed) ltp_free_host(packed);
                CHECK(rc3 == LTP_OK);
                const float us = (t1.tv_sec - t0.tv_sec) * 1e6f + (t1.tv_nsec - t0.tv_nsec) * 1e-3f;
                if (i >= 20) { best = us < best ? us : best; sum += us; }
            }
            std::printf("one 7-DoF call through ltp_plan_batch_host, %s, pow rule %s: mean %.1f us, best %.1f us\n",
                        with_rows ? "switching times + sampled trajectory" : "switching times only", variant < 2 ? "libm (default)" : "exact", sum / reps, best);
        }
        CHECK(ltp_set_pow_rule(h, LTP_POW_LIBM) == LTP_OK);
    }

    // ---- 7. one handle, several host threads: single planTrajectory-sized calls, the one-joint entry points, checkInputs and
    //         ltp_set_limits (with unchanged limits) at the same time. The library takes its two locks in one order only
    //         (host_mu before mu); every caller must get its own, unchanged result and nothing may deadlock. ----
    {
        struct Result {
            std::vector<double> t_o, t_s, d, vd, rows;
            std::vector<signed char> md;
            double treq = 0;
            int slow = 0, ln = 0, stt = 0;
            unsigned long long off[2] = {0, 0};
        };
        auto call = [&](long long pq, bool with_rows, Result& r) {
            r.t_o.assign(dof * 7, -1.0); r.t_s.assign(dof * 7, -1.0); r.d.assign(dof, -1.0); r.vd.assign(dof, -1.0); r.md.assign(dof, 9);
            ltp_records hr{r.t_o.data(), r.t_s.data(), r.d.data(), r.vd.data(), r.md.data(), &r.treq, &r.slow, &r.ln, &r.stt};
            double* packed = nullptr;
            const int rc3 = ltp_plan_batch_host(h, 1, hq[0].data() + pq * dof, hq[1].data() + pq * dof, hq[2].data() + pq * dof,
                                                hq[3].data() + pq * dof, &hr, r.off, with_rows ? &packed : nullptr);
            r.rows.clear();
            if (packed) { r.rows.assign(packed, packed + r.off[1]); ltp_free_host(packed); }
            return rc3;
        };
        auto same = [&](const Result& a, const Result& b) {
            return a.t_o == b.t_o && a.t_s == b.t_s && a.d == b.d && a.vd == b.vd && a.md == b.md && a.slow == b.slow && a.ln == b.ln &&
                   a.stt == b.stt && std::memcmp(&a.treq, &b.treq, 8) == 0 && a.off[1] == b.off[1] && a.rows.size() == b.rows.size() &&
                   (a.rows.empty() || std::memcmp(a.rows.data(), b.rows.data(), a.rows.size() * 8) == 0);
        };
        const int cases = 24;
        std::vector<Result> want(2 * cases);
        for (int i = 0; i < 2 * cases; ++i) CHECK(call((i / 2 * 131) % n, i & 1, want[i]) == LTP_OK);
        double bq = 0, bt[7] = {0, 0, 0, 0, 0, 0, 0}, bd = 0;
        CHECK(ltp_opt_braking_host(h, 2, 0.7, -1.5, &bq, bt, &bd) == LTP_OK);
        std::atomic<int> bad{0}, finished{0};
        auto planner_thread = [&](int tid) {
            Result got;
            for (int rep = 0; rep < 60; ++rep) {
                const int i = (tid * 7 + rep) % (2 * cases);
                if (call((i / 2 * 131) % n, i & 1, got) != LTP_OK || !same(got, want[i])) ++bad;
            }
            ++finished;
        };
        auto lane_thread = [&]() {
            for (int rep = 0; rep < 120; ++rep) {
                double q2 = 0, t2[7] = {0, 0, 0, 0, 0, 0, 0}, d2 = 0;
                int ok = 0;
                if (ltp_opt_braking_host(h, 2, 0.7, -1.5, &q2, t2, &d2) != LTP_OK || std::memcmp(&q2, &bq, 8) != 0 ||
                    std::memcmp(t2, bt, sizeof(bt)) != 0 || d2 != bd) ++bad;
                if (ltp_check_inputs_host(h, hq[1].data(), hq[2].data(), hq[3].data(), &ok) != LTP_OK || ok != 1) ++bad;
            }
            ++finished;
        };
        auto limits_thread = [&]() {
            for (int rep = 0; rep < 40; ++rep)
                if (ltp_set_limits(h, dof, q_min, q_max, v_max, a_max, j_max) != LTP_OK) ++bad;
            ++finished;
        };
        std::vector<std::thread> th;
        th.emplace_back(planner_thread, 0);
        th.emplace_back(planner_thread, 1);
        th.emplace_back(lane_thread);
        th.emplace_back(limits_thread);
        // a deadlock must fail the test, not hang the box: watch the threads for 60 s, then give up on them
        for (int waited = 0; finished.load() < 4 && waited < 600; ++waited) {
            timespec nap{0, 100 * 1000 * 1000};
            nanosleep(&nap, nullptr);
        }
        if (finished.load() < 4) {
            std::printf("FAILED: host threads on one handle did not finish within 60 s (deadlock)\n1 failures\n");
            std::fflush(stdout);
            _exit(1);
        }
        for (auto& t : th) t.join();
        CHECK(bad.load() == 0);
        std::printf("4 host threads on one handle (plans, one-joint calls, checkInputs, set_limits): %d wrong results\n", bad.load());
    }

    {
        timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        ltp_destroy(h);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
        std::printf("ltp_destroy: %.2f ms\n", ms);
        CHECK(ms < 100.0);
    }
    std::printf("%d failures\n", g_failures);
    return g_failures ? 1 : 0;
}
