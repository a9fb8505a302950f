// example_consumer.hip — a USER-side on-device consumer of the planner's run tables (INTEGRATION.md §2
// "writing your own consumer"). Compiled with plain hipcc against include/ltp_run_tables.hpp ONLY (no library internals, and
// deliberately with hipcc's default floating-point contraction: the header pins what must not be contracted):
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -I include -o libexample_consumer.so tests/cpp/example_consumer.hip
// The tables come from ltp_build_tables_batch (include/ltp_hip.h). Two consumers, one per access form of the header:
//   example_peak_velocity   lane = (plan, joint), straight from global memory (RunTableView + for_each_sample): the largest |v| of
//                           the joint's trajectory, the first sample where it occurs, and how many samples lie above a threshold
//                           (time above threshold = count * Ts) — a velocity-limit monitor.
//   example_box_clearance   block = plan, tables of all its joints staged in LDS (fetch_run_tables + install_run_tables, RunCursor):
//                           the closest approach of the joint VECTOR q(i) to an axis-aligned box [lo, hi] in Chebyshev distance
//                           (0 = the configuration enters the box) and the first sample where it is attained — a keep-out-zone
//                           check, which is not separable per joint.
// Neither ever sees a dense row; tests/test_gpu_consumer_hook.py checks both against the same reductions of the rows
// ltp_sample_batch writes (bit for bit: maxima, minima and counts do not depend on the order) and of the CPU oracle's rows (1e-9).
#include <hip/hip_runtime.h>

#include "ltp_run_tables.hpp"

namespace {

__global__ void __launch_bounds__(256)
k_peak_velocity(const unsigned long long* __restrict__ tables, long long lanes, double Ts, double threshold,
                double* __restrict__ max_abs_v, int* __restrict__ at_sample, long long* __restrict__ samples_above)
{
    const long long lane = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= lanes) return;
    const ltp::RunTableView t{tables, (unsigned long long)lane};
    double best = -1.0;                    // no trajectory: -1, sample -1, 0 above
    int where = -1;
    long long above = 0;
    ltp::for_each_sample(t, Ts, 0, t.traj_len(), [&](int i, double, double v, double, double) {
        const double m = __builtin_fabs(v);
        if (m > best) { best = m; where = i; }
        above += m > threshold;
    });
    max_abs_v[lane] = best;
    at_sample[lane] = where;
    samples_above[lane] = above;
}

__global__ void __launch_bounds__(ltp::kRunTableThreads)
k_box_clearance(const unsigned long long* __restrict__ tables, long long plans, int dof, double Ts, const double* __restrict__ lo,
                const double* __restrict__ hi, double* __restrict__ clearance, int* __restrict__ at_sample)
{
    __shared__ ltp::JointTable jt[ltp::kRunTableJoints];
    __shared__ double s_best[ltp::kRunTableThreads];
    __shared__ int s_where[ltp::kRunTableThreads];
    for (long long p = blockIdx.x; p < plans; p += gridDim.x) {
        const ltp::PackedTableRegs regs = ltp::fetch_run_tables(tables, (unsigned long long)p * dof, dof);
        ltp::install_run_tables(jt, dof, regs.w, Ts);          // ends with a block barrier
        const int len = jt[0].nseg > 0 ? jt[0].len : 0;        // every joint of a plan has the plan's traj_len
        double best = __builtin_huge_val();
        int where = -1;
        if (len > 0) {
            ltp::RunCursor cu[ltp::kRunTableJoints] = {ltp::RunCursor(jt[0]), ltp::RunCursor(jt[1 % ltp::kRunTableJoints]), ltp::RunCursor(jt[2]),
                                                       ltp::RunCursor(jt[3]), ltp::RunCursor(jt[4]), ltp::RunCursor(jt[5]),
                                                       ltp::RunCursor(jt[6]), ltp::RunCursor(jt[7])};
            for (int i = threadIdx.x; i < len; i += ltp::kRunTableThreads) {
                double d = 0.0;                                 // Chebyshev distance of q(i) from the box
#pragma unroll
                for (int j = 0; j < ltp::kRunTableJoints; ++j) {
                    if (j < dof) {
                        cu[j].advance(jt[j], i);
                        const double q = ltp::run_eval_q(jt[j].c[cu[j].run], i - cu[j].cur + 1);
                        const double out = __builtin_fmax(lo[j] - q, q - hi[j]);
                        d = __builtin_fmax(d, out);
                    }
                }
                if (d < best) { best = d; where = i; }          // samples in increasing order per lane: the first minimum stays
            }
        }
        s_best[threadIdx.x] = best;
        s_where[threadIdx.x] = where;
        __syncthreads();
        for (int s = ltp::kRunTableThreads / 2; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) {
                const double ob = s_best[threadIdx.x + s];
                const int ow = s_where[threadIdx.x + s];
                // the smaller distance wins, among equal distances the earlier sample: independent of the reduction order
                if (ow >= 0 && (s_where[threadIdx.x] < 0 || ob < s_best[threadIdx.x] || (ob == s_best[threadIdx.x] && ow < s_where[threadIdx.x]))) {
                    s_best[threadIdx.x] = ob;
                    s_where[threadIdx.x] = ow;
                }
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            clearance[p] = s_where[0] >= 0 ? s_best[0] : __builtin_nan("");
            at_sample[p] = s_where[0];
        }
        __syncthreads();                                        // jt and the reduction arrays are reused by the next plan
    }
}

}  // namespace

extern "C" {

int example_peak_velocity(const unsigned long long* tables, long long lanes, double Ts, double threshold, double* max_abs_v,
                          int* at_sample, long long* samples_above, void* stream)
{
    if (lanes <= 0) return 0;
    hipLaunchKernelGGL(k_peak_velocity, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, (hipStream_t)stream, tables, lanes, Ts, threshold,
                       max_abs_v, at_sample, samples_above);
    return (int)hipGetLastError();
}

int example_box_clearance(const unsigned long long* tables, long long plans, int dof, double Ts, const double* lo, const double* hi,
                          double* clearance, int* at_sample, void* stream)
{
    if (plans <= 0) return 0;
    if (dof < 1 || dof > ltp::kRunTableJoints) return -1;       // (more joints: stage them in groups of kRunTableJoints)
    const long long blocks = plans < 4096 ? plans : 4096;
    hipLaunchKernelGGL(k_box_clearance, dim3((unsigned)blocks), dim3(ltp::kRunTableThreads), 0, (hipStream_t)stream, tables, plans, dof, Ts, lo, hi,
                       clearance, at_sample);
    return (int)hipGetLastError();
}

}  // extern "C"
