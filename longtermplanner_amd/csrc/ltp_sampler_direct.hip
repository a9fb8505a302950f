// ltp_sampler_direct.hip — the sampler of VERY SHORT capped rows (first-N samples with N <= 32 or so), gfx950 (round 5).
//
// What the counters said (profiles/r05_short_rows_counters.json): at a cap of 16 / 32 samples k_sample_walk is bound by vector
// INSTRUCTION ISSUE, not by HBM — 875 / 899 vector instructions per plan for 3.6 / 7.2 KB of rows, the vector ALUs 70-80 % busy.
// Two thirds of them are overhead that rows this short cannot amortise: the streaming side spends ~230 instructions per wave pass to
// produce ONE pair of samples per lane (slot addressing, run lookup and run_coef per lane), and the single builder wave of a block
// walks every run of every joint to its last sample for planTrajectory's end-limit verdict (cc:59-61).
//
// Here a lane is a (plan, joint) row set and keeps everything in registers: it walks its runs once (for_each_run, the walk of
// k_state_at / k_end_limit), and while a run lies inside the cap it evaluates that run's stored samples itself — run_eval, the same
// function with the same run-local index as every other sampler, so the rows are bit-identical — and stores them 16 bytes at a time
// into its own four rows. No LDS, no hand-over between waves, every wave of the chip walks; the walk continues to the last sample
// for the end-limit verdict as in the other samplers. The price is the store pattern: a wave instruction writes 64 separate 16-byte
// pieces (the lanes' rows are 256 bytes apart) instead of 1 KiB contiguous, and the L2 has to merge a row's pieces before the line
// leaves — affordable while a row is a line or two (<= 32 float64 samples), a loss beyond (measured: DESIGN.md §4).
#include "ltp_sampler_lds.hpp"

namespace ltp {

template <typename T, int SEM>
__global__ void __launch_bounds__(256)
k_sample_direct(long long first, long long count, int dof, double t_sample, RowSpec rows, Limits lim, Queries in, Records rec,
                const unsigned long long* __restrict__ offsets, T* __restrict__ out, unsigned long long capacity)
{
    constexpr int NF = OutVec<T>::N;                          // samples per 16-byte slot: 2 (float64) or 4 (float32)
    typedef typename OutVec<T>::type V;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= count * dof) return;
    const long long local = idx / dof;
    const int j = (int)(idx - local * dof);
    const long long p = first + local;
    const int len = rec.traj_len[p];
    const int slen = stored_len(len, rows);
    if (slen <= 0) return;                                    // no trajectory (failed before sampling)
    const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    const unsigned long long rel = offsets[p] - offsets[first];
    if (rel + 4ull * dof * stride > capacity) {               // the tile-capacity rule of every sampler: flagged, never written
        if (j == 0) atomicOr(&rec.status[p], kStatusOverflow);
        return;
    }
    const long long ix = p * in.sq + (long long)j * in.sj;
    double q = in.q_0[ix], v = in.v_0[ix], a = in.a_0[ix];
    const int sstride = rows.stride > 1 ? rows.stride : 1;
    T* const row = out + rel + (unsigned long long)j * stride;
    const unsigned long long arr = (unsigned long long)dof * stride;
    int next = 0;                                             // next stored sample to produce
    V slot[4];                                                // the 16-byte slot under construction, per array
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int h = 0; h < NF; ++h) slot[x][h] = (T)0;
    auto flush = [&](int slot_index) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            *reinterpret_cast<V*>(row + (unsigned long long)x * arr + (unsigned long long)slot_index * NF) = slot[x];
#pragma unroll
            for (int h = 0; h < NF; ++h) slot[x][h] = (T)0;   // the tail of a row's last slot is zero (as the other samplers leave it)
        }
    };
    for_each_run<SEM>(lim, rec, p * dof + j, j, len, t_sample, q, v, a, [&](int b, int e, const RunCoef& rc) {
        // stored sample k is trajectory sample k * sstride; this run holds the trajectory samples [b, e)
        while (next < slen) {
            const int t = next * sstride;
            if (t >= e) break;
            double x4[4];
            run_eval(rc.c, t - b + 1, x4[0], x4[1], x4[2], x4[3]);
            const int h = next % NF;
#pragma unroll
            for (int x = 0; x < 4; ++x) {
#pragma unroll
                for (int hh = 0; hh < NF; ++hh)
                    if (hh == h) slot[x][hh] = (T)x4[x];
            }
            ++next;
            if (next % NF == 0) flush(next / NF - 1);
        }
        // C++ semantics: on to the last sample for the end-limit verdict; LTPlanner.m has no position limits: done with the rows, done
        return SEM == kSemMatlab && next >= slen;
    }, j == dof - 1);
    if (next % NF != 0) flush(next / NF);
    if constexpr (SEM == kSemCpp) {
        if (q < lim.q_min[j] || q > lim.q_max[j]) atomicOr(&rec.status[p], kStatusEndLimit);   // cc:59-61: the last trajectory sample
    }
}

// rows this kernel is meant for: a cap (the only way rows are known to be short before the lengths are), at most kDirectCapBytes per row
bool sample_direct_applies(int dof, RowSpec rows) { return dof >= 1 && rows.max_samples > 0; }

void launch_sample_direct(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
                          const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity, RowSpec rows, int semantics)
{
    if (count <= 0 || dof <= 0) return;
    const long long total = count * dof;
    const dim3 grid((unsigned)((total + 255) / 256)), block(256);
#define LTP_DIRECT_CASE(TY, SEM) hipLaunchKernelGGL((k_sample_direct<TY, SEM>), grid, block, 0, s, first, count, dof, t_sample, rows, lim, in, rec, offsets, (TY*)out, capacity)
    switch ((f32 ? 1 : 0) | (semantics == kSemMatlab ? 2 : 0)) {
    case 0: LTP_DIRECT_CASE(double, kSemCpp); break;
    case 1: LTP_DIRECT_CASE(float, kSemCpp); break;
    case 2: LTP_DIRECT_CASE(double, kSemMatlab); break;
    default: LTP_DIRECT_CASE(float, kSemMatlab); break;
    }
#undef LTP_DIRECT_CASE
}

}  // namespace ltp
