// ltp_sampler.hip — getTrajectory (cc:706-841) as the HBM-bound sampler k_sample, gfx950: persistent blocks that build each
// item's run tables in LDS (ltp_sampler_lds.hpp) and stream its rows. The table-pass sampler for short rows lives in
// ltp_sampler_tab.hip, the consumers and the table pass in ltp_consumers.hip, the one-launch single call in ltp_plan_small.hip.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (see Makefile). No fast-math:
// the inf/NaN flow of the reference is part of the contract.
#include "ltp_sampler_lds.hpp"

namespace ltp {

// Persistent work-queue form: as many blocks as the chip holds, each pulling (plan, joint group) items from one
// counter until it runs dry. With static round-robin dispatch the eight XCDs finish their equal shares up to 15 %
// apart (they do not write to all HBM channels at the same speed), which left a 2-4 ms tail of a 23-29 ms launch at
// reduced bandwidth; pulling keeps every XCD busy to the end. The counter sees ~20 pulls/us, far below the
// ~90/us a single word sustains. Exit: every block leaves as soon as it draws an item >= total.
// Item order: item i -> plan (i % spread) * ceil(count/spread) + i / spread (spread = 64 by default), so blocks that
// are resident together write all over the output tile: on MI355X a narrow moving write front only reaches
// ~5.2 TB/s while writes spread over a large tile reach the fill-kernel ceiling ("What bounds the sampler").
// float32 rows need 4 samples per lane in flight: they get the register budget of 4 blocks per CU (with 5 the
// compiler spills, and a spill reload waits for every outstanding load of the wave, i.e. for the look-ahead).
// The loop runs one item ahead: the next item is drawn while the tables of the current one are being built, and its
// records are requested just before the current item's rows are streamed, so that the three dependent round trips
// an item needs (queue counter, traj_len / offset, records) travel under ~10^2 row stores instead of in front of them.
// (On gfx950 a wave's loads and stores share one in-order counter: a load result consumed behind a run of stores
// waits for all of them, so an item pays one drain of its own stores either way — but only one.)
template <bool STREAMING, bool DRY, typename T>
__global__ void __launch_bounds__(kSampleThreads, (sizeof(T) == 4 ? kSampleBlocksPerCU - 1 : kSampleBlocksPerCU))
k_sample(long long first, long long count, int dof, double t_sample, Limits lim, Queries in, Records rec,
         const unsigned long long* __restrict__ offsets, T* __restrict__ out, unsigned long long capacity,
         unsigned long long* __restrict__ stamps, int spread, RowSpec rows, unsigned long long* __restrict__ next_item,
         int draw_chunk /* items per queue draw, a power of two */)
{
    __shared__ SegTable tab;
    __shared__ unsigned long long s_item;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    const long long per = (count + spread - 1) / spread;
    const unsigned long long total = (unsigned long long)per * spread * ngroups;
    const unsigned long long off0 = offsets[first];
    // Queue draws. One device-scope counter sustains ~90 atomics/us; items of short (capped) rows are drawn faster than
    // that, so a draw takes draw_chunk consecutive items (1 for whole trajectories: a long item at the end of a launch is
    // a long tail). Thread 0 keeps the chunk; chunk_i is the same in every thread.
    unsigned long long chunk_base = 0ull;
    int chunk_i = 0;
    auto draw = [&]() -> unsigned long long {      // thread 0 only
        if (chunk_i == 0) chunk_base = atomicAdd(next_item, (unsigned long long)draw_chunk);
        return chunk_base + (unsigned long long)chunk_i;
    };

    // item -> (local plan, joint group); local >= count are the holes of the interleave
    auto decode = [&](unsigned long long item, long long& local, int& j0, int& nj) {
        const int group = (int)(item % ngroups);
        const long long slot = (long long)(item / ngroups);
        local = (slot % spread) * per + slot / spread;
        j0 = group * kSampleJointGroup;
        nj = (dof - j0) < kSampleJointGroup ? (dof - j0) : kSampleJointGroup;
    };
    auto fetch = [&](unsigned long long item) {
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        const bool some = item < total && local < count;
        return fetch_item<false>(some ? first + local : -1, j0, nj, dof, lim, in, rec, offsets);
    };

    if (threadIdx.x == 0) s_item = draw();
    chunk_i = (chunk_i + 1) & (draw_chunk - 1);
    __syncthreads();
    unsigned long long item = s_item;
    ItemRegs<false> cur = fetch(item);
    __syncthreads();   // s_item may be rewritten
    while (item < total) {
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        const long long p = first + local;
        const bool lead = threadIdx.x == 0 && j0 == 0;
        // diagnostic only (stamps == nullptr in every product call): start / tables ready / end on the 100 MHz wall clock
        if (stamps && lead && local < count) stamps[3 * local] = wall_clock64();
        // the same in every lane, but loaded per lane: readfirstlane moves them (and all the row addressing derived from
        // them) into scalar registers
        const int len = __builtin_amdgcn_readfirstlane(cur.len);   // 0: hole, failed or non-finite query -> nothing to sample
        const unsigned long long off = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(cur.off >> 32)) << 32) |
                                       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cur.off);
        const unsigned long long rel = off - off0;
        const int slen = stored_len(len, rows);           // samples actually stored per row
        const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
        bool ok = len > 0;
        if (ok && rel + 4ull * dof * stride > capacity) {
            if (lead) atomicOr(&rec.status[p], kStatusOverflow);
            ok = false;
        }
        unsigned long long drawn = 0ull;
        if (threadIdx.x == 0) drawn = draw();                          // an atomic returns while the tables are built
        chunk_i = (chunk_i + 1) & (draw_chunk - 1);
        if (ok) build_run_tables(tab, p, j0, nj, len, t_sample, lim, rec, cur.pa, cur.pb);
        if (threadIdx.x == 0) s_item = drawn;
        __syncthreads();                                               // tables complete, next item known
        const unsigned long long nitem = s_item;
        const ItemRegs<false> nxt = fetch(nitem);                      // in flight while this item streams
        if (ok) {
            if (stamps && lead) stamps[3 * local + 1] = wall_clock64();   // run tables ready
            stream_rows<STREAMING, DRY, T>(tab, j0, nj, dof, slen, stride, out + rel, rows);
        }
        __syncthreads();                                               // tables and s_item are free again
        if (stamps && ok && lead) stamps[3 * local + 2] = wall_clock64();
        item = nitem;
        cur = nxt;
    }
}

// how many blocks of a persistent (work-queue) kernel the device holds at once: 0 = k_sample float64 rows,
// 1 = k_sample float32 rows, 2 = k_envelope (ltp_consumers.hip)
int sample_resident_blocks(int device, int which)
{
    if (which == 2) return envelope_resident_blocks(device);
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    hipError_t e;
    if (which == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample<true, false, float>, kSampleThreads, 0);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample<true, false, double>, kSampleThreads, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 4;
    return cus * per_cu;
}

void launch_sample(hipStream_t s, long long first, long long count, int dof, double t_sample, Limits lim, Queries in,
                   Records rec, const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity,
                   int flags, RowSpec rows, unsigned long long* next_item, int resident_blocks, unsigned long long* stamps)
{
    if (count <= 0) return;
    int spread = (flags >> 8) & 0xFFFF;
    if (spread == 0) spread = kSampleSpread;
    if ((long long)spread > count) spread = (int)count;
    const int ngroups = (dof + kSampleJointGroup - 1) / kSampleJointGroup;
    long long blocks = resident_blocks > 0 ? resident_blocks : 1536;
    if (blocks > count * ngroups) blocks = count * ngroups;
    const dim3 grid((unsigned)blocks);
    const dim3 block(kSampleThreads);
    // flags bit 0: non-temporal stores; bit 1 (diagnostic): skip the arithmetic and store sample indices, which
    // measures the ceiling of this store pattern; bits 8..23: block interleave factor (0 = default 64, 1 = plan order)
    const int draw_chunk = queue_draw_chunk(rows, f32, dof < kSampleJointGroup ? dof : kSampleJointGroup);
#define LTP_SAMPLE_CASE(ST, DR, TY) hipLaunchKernelGGL((k_sample<ST, DR, TY>), grid, block, 0, s, first, count, dof, t_sample, lim, in, rec, offsets, (TY*)out, capacity, stamps, spread, rows, next_item, draw_chunk)
    switch ((flags & 3) | (f32 ? 4 : 0)) {
    case 0: LTP_SAMPLE_CASE(false, false, double); break;
    case 1: LTP_SAMPLE_CASE(true, false, double); break;
    case 2: LTP_SAMPLE_CASE(false, true, double); break;
    case 3: LTP_SAMPLE_CASE(true, true, double); break;
    case 4: LTP_SAMPLE_CASE(false, false, float); break;
    case 5: LTP_SAMPLE_CASE(true, false, float); break;
    case 6: LTP_SAMPLE_CASE(false, true, float); break;
    default: LTP_SAMPLE_CASE(true, true, float); break;
    }
#undef LTP_SAMPLE_CASE
}

}  // namespace ltp
