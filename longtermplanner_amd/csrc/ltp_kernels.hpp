// ltp_kernels.hpp — gfx950 kernels of the batched planner (declarations + shared structs).
//
// Pipeline for one batch (reference src/long_term_planner.cc:7-63, one query = one call):
//   k_opt_fast       stage 1 (checkInputs + optSwitchTimes per (query, joint)) without the quartic
//                    sites; lanes that reach them are compacted into queue A
//   k_opt_slow       queue A, densely, with the root finder
//   k_reduce_scale   slowest-joint reduction in LDS + the two closed-form timeScaling cases;
//                    lanes that need the polynomial cases are compacted into queue B
//   k_scaling_slow   queue B, densely: all eight timeScaling cases + reset
//   k_finalize       traj_len (cc:716-719), padded row stride, per-plan output size,
//                    block sums for the offsets scan
//   k_scan_top / k_scan_apply   exclusive scan -> packed trajectory offsets
//   k_sample         getTrajectory (cc:706-841) + end-limit check (cc:59-61), HBM-write bound
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/ltp_run_tables.hpp"   // kMaxSegments, the packed run-table format (public)

namespace ltp {

// status bits (planTrajectory's bool == (status == 0))
enum : int {
    kStatusInvalidInput = 1,   // checkInputs false (cc:14-15)
    kStatusOptFailed = 2,      // optSwitchTimes false for a joint (cc:29)
    kStatusNoSlowest = 4,      // slowest_joint == -1 (cc:39)
    kStatusEndLimit = 8,       // last q sample outside [q_min,q_max] (cc:59-61); trajectory is filled
    kStatusNonFinite = 16,     // DEFINED: non-finite switching times -> traj_len 0 (reference: UB)
    kStatusOverflow = 32,      // trajectory does not fit the caller's output tile; not sampled
    kStatusGoalOutside = 64,   // NEW, opt-in: q_goal outside [q_min,q_max]; rejected before planning (reference: unchecked)
    kStatusMatlabError = 128,  // MATLAB semantics only: LTPlanner.m would have raised an error; rejected, traj_len 0
    kStatusMatlabComplex = 256,// MATLAB semantics only, informational: LTPlanner.m would have carried a complex intermediate
                               // value; the plan continues with the real part and IS delivered
};

constexpr int kQueriesPerBlock = 64;   // one wave = 64 queries of one joint
constexpr int kMaxJointSlots = 8;      // blockDim.y of k_switch_times
constexpr int kRowAlign = 32;          // trajectory rows padded to 32 elements (256 B of doubles, 128 B of floats)
constexpr int kSampleJointGroup = 8;   // joints handled by one k_sample block
constexpr int kSampleThreads = 256;
#ifndef LTP_SAMPLE_BLOCKS_PER_CU
#define LTP_SAMPLE_BLOCKS_PER_CU 5
#endif
constexpr int kSampleBlocksPerCU = LTP_SAMPLE_BLOCKS_PER_CU;   // register budget of k_sample (512 / this many VGPRs); measured best of 4..7
constexpr int kSampleSpread = 64;      // default block->plan interleave of k_sample
constexpr int kScanBlock = 1024;       // plans per finalize/scan block

// which samples of a trajectory are stored in its rows
struct RowSpec {
    int max_samples;   // at most this many stored samples per row; 0 = no cap
    int stride;        // every stride-th sample (0, stride, 2*stride, ...); <= 1 = every sample
};

struct Limits {            // device pointers, [dof] each
    const double* q_min;
    const double* q_max;
    const double* v_max;
    const double* a_max;
    const double* j_max;
    const double* pw;      // [dof][kLimPowN]: LimPow of every joint under the handle's pow rule (k_limit_powers)
};
constexpr int kLimPowN = 7;

struct Queries {           // element (query p, joint j) at ptr[p * sq + j * sj]
    const double* q_goal;
    const double* q_0;
    const double* v_0;
    const double* a_0;
    long long sq, sj;
};

struct Records {           // query-major outputs of stages 1-3
    double* t_opt;         // [n][dof][7]
    double* t_scaled;      // [n][dof][7]
    double* dir;           // [n][dof]
    double* v_drive;       // [n][dof]
    signed char* mod;      // [n][dof]
    double* t_required;    // [n]
    int* slowest;          // [n]
    int* traj_len;         // [n]
    int* status;           // [n]
};

// items per work-queue draw of the persistent samplers: 1 for whole trajectories; for capped rows as many as keep a draw at
// >= ~256 KB of rows, at most 8
inline int queue_draw_chunk(RowSpec rows, bool f32, int joints_per_item)
{
    if (rows.max_samples <= 0) return 1;
    const long long item_bytes = 4ll * (f32 ? 4 : 8) * rows.max_samples * joints_per_item;
    int k = 1;
    while (k < 8 && item_bytes * (2 * k) <= 262144) k *= 2;
    return k;
}

// variant = semantics (bit 0: 0 = the C++ reference, 1 = LTPlanner.m) | kPowLibm (bit 1: the pow rule LTP_POW_LIBM); the template
// parameter SEM of ltp_profile.hpp and of the stage kernels. f is called with std::integral_constant<int, variant>.
template <class F>
inline void dispatch_variant(int variant, F&& f)
{
    switch (variant & 3) {
    case 0: f(std::integral_constant<int, 0>{}); break;
    case 1: f(std::integral_constant<int, 1>{}); break;
    case 2: f(std::integral_constant<int, 2>{}); break;
    default: f(std::integral_constant<int, 3>{}); break;
    }
}

long long queue_segment(long long n, int dof);   // entries per queue shard; a batch needs 2 * 8 * this many u64
void launch_switch_times(hipStream_t s, long long n, int dof, double t_sample, int goal_check, Limits lim, Queries in,
                         Records out, signed char* lane_flags, unsigned long long* queue_items, unsigned long long* counts,
                         int variant = 0 /* semantics | pow rule << 1 (dispatch_variant) */);
void launch_offsets(hipStream_t s, long long n, int dof, double t_sample, Records rec,
                    unsigned long long* block_sums, unsigned long long* offsets, bool lens_ready, RowSpec rows);
// Run tables: built inside the sampler / envelope kernel by the item's block, or by the table pass —
// launch_build_tables(first, count, ...) leaves table_bytes(count * dof) bytes in `tables` (912 bytes per plan and joint: the
// packed form, which the consumer expands with run_coef()) for launch_sample_tab / launch_envelope(tables != nullptr). base_first: the plan whose offset is the origin of out / env (== first unless a
// range is processed in pieces that share one table buffer).
unsigned long long table_bytes(long long lanes /* plans * dof */);
void launch_build_tables(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
                         RowSpec rows, bool whole_trajectory /* false: only the runs capped rows touch */,
                         const unsigned long long* offsets /* or nullptr */, long long base_first /* row offsets relative to this plan */,
                         unsigned long long* tables, int semantics = 0);
void launch_sample(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in,
                   Records rec, const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity,
                   int flags, RowSpec rows, unsigned long long* next_item /* zeroed on the same stream */,
                   int resident_blocks, unsigned long long* stamps = nullptr);
void launch_sample_tab(hipStream_t s, long long first, long long count, long long base_first, int dof, Records rec,
                       const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity, int flags, RowSpec rows,
                       unsigned long long* next_item /* zeroed on the same stream */, int resident_blocks, const unsigned long long* tables,
                       double t_sample /* the one the tables were built with */,
                       unsigned long long* stamps = nullptr /* diagnostic: 8 per (plan, joint group) item */);
// Rows without any table traffic (every row format, both semantics, any number of joints; taken by itself for capped, float32 and
// sparse rows and in MATLAB semantics): a builder wave per block walks the runs into LDS, five streaming waves write the rows
// (ltp_sampler_walk.hip).
bool sample_walk_applies(int dof, RowSpec rows);
int sample_walk_resident_blocks(int device, bool f32);
// returns true if the autonomous-wave form took the rows (caps of at most 32 samples; flags bit 7 forbids it)
bool launch_sample_walk(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
                        const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity, int flags, RowSpec rows,
                        unsigned long long* next_item /* zeroed on the same stream */, int resident_blocks, int semantics = kSemCpp,
                        int auto_cus = 0 /* sample_walk_auto_prepare(device) */);
// per device, once, outside stream capture: dynamic-LDS limit of the autonomous-wave kernels (checked) -> compute units (0: *err)
int sample_walk_auto_prepare(int device, hipError_t* err);
int sample_tab_resident_blocks(int device, bool f32);
int sample_resident_blocks(int device, int which /* 0 k_sample f64, 1 k_sample f32, 2 k_envelope */);
int envelope_resident_blocks(int device);
// the analytic envelopes by a lane-per-(plan, joint) register walk: no run tables, no workspace (ltp_consumers.hip: k_envelope_walk)
void launch_envelope_walk(hipStream_t s, long long first, long long count, long long base_first, int dof, double t_sample, Limits lim, Queries in,
                          Records rec, int window, int n_windows, double* env, int semantics);
void launch_envelope(hipStream_t s, long long first, long long count, long long base_first, int dof, double t_sample, Limits lim, Queries in,
                     Records rec, int window, int n_windows, double* env, unsigned long long* next_item /* zeroed on the same stream */,
                     int resident_blocks, unsigned long long* probe = nullptr /* diagnostic: 16 stamps per item */,
                     const unsigned long long* tables = nullptr,
                     bool analytic = false /* extreme samples from the roots of q'(m) per run instead of every sample: 1e-15, not bit-identical */);
void launch_replan_states(hipStream_t s, long long first, long long count, int dof, RowSpec rows, Queries in, Records rec,
                          const unsigned long long* offsets, const void* tile, bool f32, unsigned long long capacity,
                          const int* sample_index, int uniform_index,
                          double* q_0, double* v_0, double* a_0, long long sq, long long sj,
                          double t_sample, Limits lim, int semantics /* float64 tiles: the states are recomputed from the records, same bits */);
void launch_end_limit(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec);
void launch_state_at(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in,
                     Records rec, const int* sample_index, int uniform_index, double* q_0, double* v_0, double* a_0,
                     long long sq, long long sj, int semantics = 0);
// planTrajectory for n queries with n * dof <= small_batch_pairs() in one launch of one block; every pointer may be host
// memory the device can address (pinned). rows == nullptr: no sampling (status still carries the end-limit verdict).
// *done becomes 1 when all results are visible to the host, 2 if the rows did not fit `capacity` (then nothing was sampled).
// The grid has small_batch_blocks() blocks (one per joint when rows are written); end_flags: [blocks][n] ints the host ORs into
// status; arrivals: one zeroed device word.
int small_batch_pairs();
int small_batch_blocks(int dof, bool with_rows);
void launch_plan_small(hipStream_t s, int n, int dof, double t_sample, int goal_check, RowSpec rows, Limits lim, const double* const in[4],
                       Records rec, unsigned long long* offsets, double* out_rows, unsigned long long capacity, int* end_flags,
                       unsigned int* arrivals, volatile int* done,
                       bool records_given = false /* getTrajectory: t_scaled, dir, mod, v_drive of `rec` are inputs */,
                       bool libm_pow = false /* the pow rule LTP_POW_LIBM */,
                       bool host_inputs = false /* in[] is host memory this thread may read now: a few queries then ride in the kernel arguments */);
void launch_generate(hipStream_t s, long long n, int dof, Limits lim, unsigned long long seed, long long first_query,
                     double* q_goal, double* q_0, double* v_0, double* a_0, long long sq, long long sj);

// LimPow tables of dof joints under both pow rules ([dof][kLimPowN] each), from the device copies of a_max / j_max
void launch_limit_powers(hipStream_t s, int dof, const double* a_max, const double* j_max, double* out_exact, double* out_libm);
void launch_check_inputs(hipStream_t s, int dof, Limits lim, const double* q_0, const double* v_0, const double* a_0, int* ok, int variant = 0);
// single-joint mirrors of the protected methods (one lane); io[11] receives the lane's MATLAB flags (kMatlabComplex | kMatlabError)
void launch_single_opt_braking(hipStream_t s, int joint, double t_sample, Limits lim, double v_0, double a_0, double* out10, int variant = 0);
void launch_single_opt_switch(hipStream_t s, int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0,
                              double a_0, double v_drive, double* io10, int variant = 0);
void launch_single_time_scaling(hipStream_t s, int joint, double t_sample, Limits lim, double q_goal, double q_0, double v_0,
                                double a_0, double dir, double t_required, double* out11, int variant = 0);
// MATLAB's roots() on the device (ltp_roots_matlab.hpp): n polynomials of `degree` <= 6, [n][degree+1] coefficients; re, im
// [n][degree] in MATLAB's output order, nroots [n], status [n] (0 ok, 1 no convergence, 2 NaN / Inf)
void launch_roots_matlab(hipStream_t s, long long n, int degree, const double* coef, double* re, double* im, int* nroots, int* status);
void launch_math_probe(hipStream_t s, long long n, const double* x, const double* y, double* out);
void launch_libm_pow_probe(hipStream_t s, long long n, const double* x, const double* y, double* out);   // out[i] = the restated glibc pow(x[i], y[i])
void launch_roots_probe(hipStream_t s, long long n, int degree, const double* coef, double* root);
void launch_roots_all(hipStream_t s, long long n, int degree, bool f32, const void* coef, void* re, void* im);

}  // namespace ltp
