// ltp_sampler_tab.hip — the sampler for rows that are short compared with an item's fixed costs (first-N-samples rows,
// receding-horizon rows), gfx950: run tables from the table pass (k_build_tables, ltp_consumers.hip), a loader wave with LDS-direct
// loads, seven streaming waves that never wait. Rows are bit-identical to k_sample's (ltp_sampler.hip).
// Since round 4 nothing takes this sampler by itself (ltp_sampler_walk.hip keeps the tables in the compute unit and is ahead at every
// cap); it stays ON REQUEST (ltp_sample_batch flag bit 2) as the library's own reader of the packed table format that
// include/ltp_run_tables.hpp publishes, and as the A/B partner of bench.py --no-walk.
#include "ltp_sampler_lds.hpp"

namespace ltp {

// ---------------------------------------------------------------------------------------
// The sampler for rows that are short compared with an item's fixed costs (first-N-samples rows, receding-horizon rows).
// k_sample pays per item: a table build of ~8 us of latency, three block barriers, and — because a wave's loads and stores
// share one in-order counter on gfx950 — one complete drain of the wave's own row stores before it can consume the next
// item's prefetched records. For 1.7 k-sample rows other blocks of the CU cover that; for a few hundred samples per row it
// is most of the item (measured: 3.6-3.8 TB/s for 256-sample rows, the same with and without the table build).
// Here the roles are split between the waves of a block:
//   * the run tables come from the table pass (k_build_tables), compact: only the runs the stored samples touch;
//   * the last wave, the loader, draws the coming items and brings their tables into one of kTabBuffers LDS buffers with
//     LDS-direct loads, two items ahead; it issues loads but never row stores;
//   * the other waves, the streaming waves, each own one joint of the current item and write its rows from the item's
//     buffer, slots with a run boundary evaluated in place; they issue stores but never loads, so nothing they execute ever
//     waits for a store to complete;
//   * buffers change hands through LDS flags (s_ready / s_consumed in sample_tab_body), not block barriers: a fast wave
//     runs up to kTabBuffers - 1 items ahead of a slow one.
// Rows are bit-identical to k_sample's: same tables (for_each_run == the cooperative build), same per-sample arithmetic.
// ---------------------------------------------------------------------------------------
constexpr int kTabStreamWaves = 7;                              // streaming waves per block; wave kTabStreamWaves is the loader
constexpr int kTabThreads = (kTabStreamWaves + 1) * 64;          // 8 waves: two per SIMD
constexpr int kTabJointGroup = 7;                                // joints per item: one row set per streaming wave
constexpr int kTabBuffers = 4;                                   // LDS table buffers per block (11.9 KB each)
struct TabItem {
    unsigned long long rel;               // element offset of the plan inside `out`
    int slen;                             // stored samples per row; 0 = nothing to stream (hole, failed plan, does not fit)
    int j0, nj;
    int done;                             // 1 = the queue is exhausted
    unsigned long long item;              // queue position (diagnostic stamps only)
};
struct alignas(16) TabBuffer {
    JointTable jt[kTabJointGroup];        // filled by LDS-direct loads (16 bytes per lane) from the table pass's output
    TabItem hdr;                          // what the streaming waves read, written by the loader when the loads are in
};
static_assert(sizeof(JointTable) % 16 == 0, "LDS-direct loads land 16 bytes per lane");

// What a streaming wave does with an item: wave w owns joint w (nj <= 3: several waves share a joint) and writes the
// joint's four rows, 64 slots = 1 KiB per row and step. Same slot arithmetic as stream_rows' pass A, except that a slot which
// contains a run boundary (or the end of the row) is evaluated sample by sample in place, by exactly the steps of
// stream_rows' pass B, instead of being picked up from LDS: rows this kernel is meant for are one or two wave steps long, and a
// pass B costs the wave ~2 us per item in which it issues no store.
template <bool STREAMING, typename T>
LTP_DEV void tab_stream(const TabBuffer& B, const TabItem& hdr /* B.hdr, already in registers */, int dof, T* __restrict__ out, int sstride, int wave)
{
    typedef typename OutVec<T>::type V;
    constexpr int N = OutVec<T>::N;
    // the lane id is recomputed per item: kept in a register across the kernel it ends up spilled (the loader branch needs
    // the registers), and a scratch reload here would wait for every row store the wave has in flight
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int slen = hdr.slen, j0 = hdr.j0, nj = hdr.nj;
    if (slen <= 0) return;
    const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
    const unsigned long long arr_stride = (unsigned long long)dof * stride;
    const int nslots = (slen + N - 1) / N;
    // Lanes per joint: a row of at most 32 (16) slots leaves half (three quarters) of a wave without a slot, and what a
    // streaming wave costs is the instructions it issues, not the lanes that execute them: such rows share a wave between
    // two (four) joints — 64-sample rows of a 7-joint item take four waves' worth of instructions instead of seven.
    const int lg = nslots > 32 ? 6 : (nslots > 16 ? 5 : 4);
    int jl2, first_slot, step;
    if (lg == 6) {
        const int wpr = nj >= 4 ? 1 : (nj == 3 ? 2 : (nj == 2 ? 3 : 7));     // waves per joint
        jl2 = wpr == 1 ? wave : (wpr == 2 ? wave >> 1 : (wpr == 3 ? (wave >= 3) + (wave >= 6) : 0));   // wave / wpr without a division
        first_slot = (wave - jl2 * wpr) * 64 + lane;
        step = 64 * wpr;
    } else {
        jl2 = (wave << (6 - lg)) + (lane >> lg);
        first_slot = lane & ((1 << lg) - 1);
        step = 64;                                                         // (one slot per lane)
    }
    const bool mine = jl2 < nj;
    if (__builtin_amdgcn_ballot_w64(mine) == 0ull) return;
    const JointTable& jt = B.jt[mine ? jl2 : 0];
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    // one buffer descriptor over the item's rows (all four arrays of all its joints), a 32-bit offset per lane; items beyond
    // 4 GB of rows (trajectories of millions of samples) take ordinary stores
    const unsigned long long item_bytes = 4ull * arr_stride * sizeof(T);
    const bool buffer_stores = STREAMING && sizeof(T) == 8 && item_bytes <= 0xffffff00ull;
    T* const item = out + hdr.rel;
    const unsigned long long row_at = (unsigned long long)(j0 + (mine ? jl2 : 0)) * stride;     // element offset of the joint's q row
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(item, 0, buffer_stores ? (int)(unsigned)item_bytes : 0, 0x00020000);
    const int* st = jt.start;
    const int nruns = jt.nseg;
    int kr = 0, cur = 0, nxt = nruns > 1 ? st[1] : 0x7fffffff;
    for (int slot = mine ? first_slot : nslots; slot < nslots; slot += step) {
        const int i0 = N * slot;
        V o[4];
        const int t0 = i0 * sstride;
        while (nxt <= t0) {
            ++kr;
            cur = nxt;
            nxt = kr + 1 < nruns ? st[kr + 1] : 0x7fffffff;
        }
        const bool straddles = t0 + (N - 1) * sstride >= nxt;
        if (straddles || i0 + N > slen) {
            // a run boundary or the end of the row inside the slot: sample by sample (the tail of the last slot is
            // row padding and stays zero)
            int kh = kr, ch = cur, nh = nxt;
#pragma unroll
            for (int h = 0; h < N; ++h) {
                const int i = t0 + h * sstride;
                while (nh <= i) {
                    ++kh;
                    ch = nh;
                    nh = kh + 1 < nruns ? st[kh + 1] : 0x7fffffff;
                }
                const bool pad = i0 + h >= slen;
                double x4[4];
                run_eval(jt.c[kh], i - ch + 1, x4[0], x4[1], x4[2], x4[3]);
#pragma unroll
                for (int x = 0; x < 4; ++x) o[x][h] = pad ? (T)0 : (T)x4[x];
            }
        } else {
            double c[kRunCoefs];
#pragma unroll
            for (int x = 0; x < kRunCoefs; ++x) c[x] = jt.c[kr][x];
#pragma unroll
            for (int h = 0; h < N; ++h) {
                double x4[4];
                run_eval(c, t0 + h * sstride - cur + 1, x4[0], x4[1], x4[2], x4[3]);
#pragma unroll
                for (int x = 0; x < 4; ++x) o[x][h] = (T)x4[x];
            }
        }
        if (buffer_stores) {
            if constexpr (STREAMING && sizeof(T) == 8) {
                const unsigned voff = (unsigned)((row_at + (unsigned long long)i0) * sizeof(T));
                const unsigned arr_bytes = (unsigned)(arr_stride * sizeof(T));
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o[x]), rsrc, voff + (unsigned)x * arr_bytes, 0, /*nt | sc1*/ 2 | 16);
            }
        } else if constexpr (STREAMING) {
#pragma unroll
            for (int x = 0; x < 4; ++x) __builtin_nontemporal_store(o[x], reinterpret_cast<V*>(item + row_at + x * arr_stride + i0));
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) *reinterpret_cast<V*>(item + row_at + x * arr_stride + i0) = o[x];
        }
    }
}

// The lane id, recomputed (two instructions) and opaque to common-subexpression elimination: kept live across the loader's
// loop it gets spilled, and a scratch reload in that loop waits for the table loads in flight (~8 us each time).
LTP_DEV int fresh_lane()
{
    int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
}

// s_waitcnt with only the vector-memory counter set (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[5:4] << 14)
#define LTP_WAIT_VMCNT(N) __builtin_amdgcn_s_waitcnt((((N) & 15) | (((N) >> 4) << 14)) | (7 << 4) | (15 << 8))

// LDS accesses of the loader wave, as instructions the compiler does not model. Once a wave has LDS-direct loads in flight
// the compiler puts "wait for ALL vector-memory operations" in front of every LDS access it knows of (it cannot tell that the
// loads land elsewhere), which would drain the prefetches at every flag poll and header write. These are plain ds_read /
// ds_write: LDS serves a wave's requests in order, reads wait for their own data, and the loader orders them against the
// LDS-direct loads itself (LTP_WAIT_VMCNT). Untracked LDS operations can only make the compiler's own lgkmcnt waits stricter.
LTP_DEV unsigned lds_offset(const void* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; }
LTP_DEV int lds_peek32(unsigned a)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
// several reads, one wait: an LDS round trip is ~150 cycles in a CU full of streaming waves
LTP_DEV void lds_peek64x4(unsigned a0, unsigned a1, unsigned a2, unsigned a3, unsigned long long (&v)[4])
{
    asm volatile("ds_read_b64 %0, %4\n\tds_read_b64 %1, %5\n\tds_read_b64 %2, %6\n\tds_read_b64 %3, %7\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
}
LTP_DEV void lds_peek64x5(unsigned a, unsigned long long (&v)[5])      // five consecutive 8-byte words
{
    asm volatile("ds_read_b64 %0, %5\n\tds_read_b64 %1, %5 offset:8\n\tds_read_b64 %2, %5 offset:16\n\tds_read_b64 %3, %5 offset:24\n\t"
                 "ds_read_b64 %4, %5 offset:32\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]) : "v"(a) : "memory");
}
LTP_DEV void lds_poke32(unsigned a, int v) { asm volatile("ds_write_b32 %0, %1" : : "v"(a), "v"(v) : "memory"); }
LTP_DEV void lds_poke64(unsigned a, unsigned long long v) { asm volatile("ds_write_b64 %0, %1" : : "v"(a), "v"(v) : "memory"); }

// Loader wave: turns the packed tables that LDS-direct loads left at the end of the item's JointTables into the tables the
// streaming waves read (header words to the front, ten coefficients per run from its five packed words, by run_coef()).
// Lane -> (joint lane / 9, run 9 * pass + lane % 9): 63 lanes per pass, one pass for capped rows, three for whole tables.
// Order matters, because the expanded words overwrite the packed ones: a pass writes JointTable words 12 + 90 k .. 101 + 90 k,
// the packed header sits in words 98 .. 111 and the packed run r in words 112 + 5 r .. 116 + 5 r — so the header (and vsnap) is
// read before pass 0 writes, every lane of a pass reads its run before any lane writes (the LDS serves a wave in order and
// lds_peek waits for its data), and what pass k overwrites is below what the later passes still have to read.
// The same expansion by the streaming wave that owns the joint (ordinary LDS accesses: a streaming wave has no LDS-direct loads
// in flight). Lane r < nseg expands run r, lanes 32..43 move the header; every lane has read before any lane writes (one wave,
// one instruction stream, and the LDS serves it in order).
template <class Buffer>
LTP_DEV void expand_packed_tables(Buffer& B, int nj, int max_runs, int lane, double Ts)
{
    constexpr int kRunsPerPass = 9;
    static_assert(kTabJointGroup * kRunsPerPass <= 64, "a pass is one wave");
    static_assert(12 + kRunCoefs * kRunsPerPass <= kPackedAt + kPackedHeaderWords - 2, "pass 0 stays below vsnap and the packed runs");
    static_assert(12 + kRunCoefs * 2 * kRunsPerPass <= kPackedAt + kPackedHeaderWords + kPackedRunWords * 2 * kRunsPerPass && kMaxSegments <= 3 * kRunsPerPass,
                  "pass 1 stays below the packed runs of pass 2, and there is no pass 3");
    const int x = lane / kRunsPerPass, i = lane - x * kRunsPerPass;
    const bool joint = x < nj;
    const unsigned jt = lds_offset(&B.jt[joint ? x : 0]);
    const unsigned pk = jt + (unsigned)kPackedAt * 8u;
    unsigned long long hd[4];
    lds_peek64x4(pk + 8u * (unsigned)i, pk + 8u * (unsigned)(9 + (i < 3 ? i : 0)), pk, pk + 12u * 8u, hd);
    const int nseg = (int)(unsigned)hd[2];
    const double vsnap = __builtin_bit_cast(double, hd[3]);
    if (joint) {
        lds_poke64(jt + 8u * (unsigned)i, hd[0]);
        if (i < 3) lds_poke64(jt + 8u * (unsigned)(9 + i), hd[1]);
    }
    for (int r0 = 0; r0 < max_runs; r0 += kRunsPerPass) {
        const int r = r0 + i;
        const bool live = joint && r < nseg;
        const unsigned src = pk + (unsigned)(kPackedHeaderWords + kPackedRunWords * (live ? r : 0)) * 8u;
        unsigned long long st[5];
        lds_peek64x5(src, st);
        const RunCoef rc = run_coef<kSemMatlab>((int)(unsigned)st[4], __builtin_bit_cast(double, st[3]), __builtin_bit_cast(double, st[0]),
                                                __builtin_bit_cast(double, st[1]), __builtin_bit_cast(double, st[2]), vsnap, Ts);   // (a superset of the C++ modes: same bits)
        if (live) {
            const unsigned dst = jt + (unsigned)(12 + kRunCoefs * r) * 8u;
#pragma unroll
            for (int c = 0; c < kRunCoefs; ++c) lds_poke64(dst + 8u * (unsigned)c, __builtin_bit_cast(unsigned long long, rc.c[c]));
        }
    }
}

template <bool STREAMING, typename T>
LTP_DEV void sample_tab_body(long long first, long long count, long long base_first, int dof, Records rec,
                             const unsigned long long* __restrict__ offsets, T* __restrict__ out, unsigned long long capacity, int spread,
                             RowSpec rows, unsigned long long* __restrict__ next_item, const unsigned long long* __restrict__ tables,
                             int draw_chunk, unsigned long long* __restrict__ stamps /* diagnostic: 8 per item, nullptr in product calls */,
                             double t_sample)
{
    // stamps[8 * item + k] (wall clock, tools/tab_probe.py): loader — 0 its iteration starts (a buffer is free), 6 the next
    // item's loads are issued, 1 this item's loads are in, 3 it is published, 7 the previous publication, 2 = 1 if the tables
    // needed the second fetch; streaming wave 0 — 4 it starts the item, 5 its rows are issued.
    // Hand-over of the table buffers without block barriers: buffer s % kTabBuffers holds the block's s-th item once the
    // loader has set s_ready[s % kTabBuffers] = s + 1; streaming wave w has finished s_consumed[w] items. The loader reuses
    // a buffer when every wave is past the item that was in it; a fast wave may thus run kTabBuffers - 1 items ahead of a
    // slow one (with one barrier per item every wave waited for the slowest: 3.1 of 8.8 us at 256-sample rows). All eight
    // waves of a block are resident together, the loader waits only for the streaming waves and they only for the loader, the
    // loader publishes a final "done" item and every wave leaves on reading it: no wait can last forever.
    __shared__ TabBuffer buf[kTabBuffers];
    __shared__ int s_ready[kTabBuffers];
    __shared__ int s_consumed[kTabStreamWaves];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int sstride = rows.stride > 1 ? rows.stride : 1;
    if (threadIdx.x < kTabBuffers) s_ready[threadIdx.x] = 0;
    if (threadIdx.x < kTabStreamWaves) s_consumed[threadIdx.x] = 0;
    __syncthreads();
    if (wave < kTabStreamWaves) {
        // ---- streaming waves: LDS reads and row stores only ----
        for (int seq = 0;; ++seq) {
            const int b = seq % kTabBuffers;
            while (__hip_atomic_load(&s_ready[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != seq + 1) __builtin_amdgcn_s_sleep(1);   // (longer sleeps: no gain)
            // the header in one LDS round trip (two 16-byte reads), wave-uniform
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            static_assert(sizeof(TabItem) == 32 && offsetof(TabItem, slen) == 8 && offsetof(TabItem, nj) == 16 && offsetof(TabBuffer, hdr) % 16 == 0, "read as two u32x4");
            const u32x4 h0 = reinterpret_cast<const u32x4*>(&buf[b].hdr)[0], h1 = reinterpret_cast<const u32x4*>(&buf[b].hdr)[1];
            if (__builtin_amdgcn_readfirstlane((int)h1[1])) break;                       // done
            TabItem hdr;
            hdr.rel = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)h0[1]) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)h0[0]);
            hdr.slen = __builtin_amdgcn_readfirstlane((int)h0[2]);
            hdr.j0 = __builtin_amdgcn_readfirstlane((int)h0[3]);
            hdr.nj = __builtin_amdgcn_readfirstlane((int)h1[0]);
            const bool stamp = stamps && wave == 0 && (threadIdx.x & 63) == 0;
            const unsigned long long it = stamp ? ((unsigned long long)h1[3] << 32) | h1[2] : 0ull;
            if (stamp) stamps[8 * it + 4] = wall_clock64();
            tab_stream<STREAMING, T>(buf[b], hdr, dof, out, sstride, wave);
            if (stamp) stamps[8 * it + 5] = wall_clock64();
            // the wave's reads of buf[b] are complete (release orders its LDS traffic; row stores need not be: they carry
            // their data in registers)
            if ((threadIdx.x & 63) == 0) __hip_atomic_store(&s_consumed[wave], seq + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return;
    }
    // ---- loader wave: loads only. Under the sampler's own write traffic a global read takes ~8 us, longer than an item
    // streams, so an item's data is requested kTabAhead items before it is published — and it is requested with LDS-DIRECT
    // loads (buffer_load ... lds, 16 bytes per lane straight into the JointTable of the item's buffer): no registers are held,
    // and because such a load has no register result the compiler inserts no wait of its own for it; the one wait is written
    // here. Loads complete in order, and every item issues exactly kPerItem of them (holes and the items after the end of the
    // queue re-read item 0's tables into a buffer nobody streams), so "item seq is in" is "at most kTabAhead * kPerItem
    // vector-memory operations outstanding" — a constant. Anything else the loader issues in between (queue draws, status
    // bits, stamps) only makes that wait stricter. The loader shares its SIMD with five streaming waves that keep the vector
    // ALU busy: it runs at raised issue priority. ----
    __builtin_amdgcn_s_setprio(3);
    constexpr int kTabAhead = 2;
    constexpr int kCappedRuns = 8;
    constexpr int kCappedPairs = (kPackedHeaderWords + kCappedRuns * kPackedRunWords) / 2;   // 27 word pairs: header + 8 packed runs
    constexpr unsigned kPackedByte = (unsigned)kPackedAt * 8u;                               // where the packed form lands in a JointTable
    static_assert(kTabAhead + 2 <= kTabBuffers, "buffers: one being streamed, one being published, kTabAhead in flight");
    const int ngroups = (dof + kTabJointGroup - 1) / kTabJointGroup;
    const long long per = (count + spread - 1) / spread;
    const unsigned long long total = (unsigned long long)per * spread * ngroups;
    constexpr unsigned kTileBytes = (unsigned)kPackedWords * 64u * 8u;
    // capped rows mostly touch the first runs only (a switch of the jerk profile cuts up to three runs: 8 runs is what
    // 256 samples of a 7-DoF plan need in 99.95 % of the items): kCappedPairs word pairs (54 words, 8 packed runs) per joint — an
    // item's joints are neighbours in the table tile, so its seven loads fetch 27 lines of 128 bytes; whole tables are 57 pairs
    const bool whole_tables = rows.max_samples <= 0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto uniform64 = [](unsigned long long x) -> unsigned long long {
        return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(x >> 32)) << 32) |
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)x);
    };
    // Queue positions: a draw takes draw_chunk consecutive items (one device-scope counter sustains ~90 atomics/us; short
    // items are drawn faster than that). The atomic of the next chunk is issued at the top of the iteration that hands out
    // the current chunk's last position and consumed at the bottom of the SAME iteration: in between lies straight-line code,
    // so the compiler can count what was issued after it and waits for the atomic alone — carried across the loop it would
    // wait for everything outstanding, i.e. drain the prefetches.
    unsigned long long chunk_cur = 0ull, chunk_pending = 0ull;
    int chunk_i = 0;
    bool chunk_wanted = false;
    if (fresh_lane() == 0) chunk_pending = atomicAdd(next_item, (unsigned long long)draw_chunk);
    chunk_cur = uniform64(chunk_pending);
    auto next_item_id = [&]() __attribute__((always_inline)) -> unsigned long long {       // top of an iteration
        const unsigned long long id = chunk_cur + (unsigned long long)chunk_i;
        if (++chunk_i == draw_chunk) {
            if (fresh_lane() == 0) chunk_pending = atomicAdd(next_item, (unsigned long long)draw_chunk);
            chunk_wanted = true;
        }
        return id;
    };
    auto finish_draw = [&]() __attribute__((always_inline)) {                               // bottom of the same iteration
        __builtin_amdgcn_sched_barrier(0);              // not to be hoisted in front of the iteration's loads (it would wait there)
        if (chunk_wanted) {
            chunk_cur = uniform64(chunk_pending);
            chunk_i = 0;
            chunk_wanted = false;
        }
    };
    // item -> (plan inside [first, first + count) or -1, first joint, joints). 64-bit divisions are ~10^2 instructions each on
    // this machine and the loader pays them per item: one joint group (dof <= 7) and a power-of-two interleave (the default,
    // 64) need none, anything else that fits 32 bits uses 32-bit division.
    const int spread_log2 = (spread & (spread - 1)) == 0 ? 31 - __builtin_clz((unsigned)spread) : -1;
    auto decode = [&](unsigned long long item, long long& local, int& j0, int& nj) __attribute__((always_inline)) {
        local = -1; j0 = 0; nj = 0;
        if (item >= total) return;
        unsigned long long slot = item;
        int group = 0;
        if (ngroups > 1) {
            if (total <= 0xffffffffull) { group = (int)((unsigned)item % (unsigned)ngroups); slot = (unsigned)item / (unsigned)ngroups; }
            else { group = (int)(item % (unsigned long long)ngroups); slot = item / (unsigned long long)ngroups; }
        }
        long long l;
        if (spread_log2 >= 0) l = (long long)(slot & (unsigned long long)(spread - 1)) * per + (long long)(slot >> spread_log2);
        else if (total <= 0xffffffffull) l = (long long)((unsigned)slot % (unsigned)spread) * per + (long long)((unsigned)slot / (unsigned)spread);
        else l = (long long)(slot % (unsigned long long)spread) * per + (long long)(slot / (unsigned long long)spread);
        j0 = group * kTabJointGroup;
        nj = (dof - j0) < kTabJointGroup ? (dof - j0) : kTabJointGroup;
        if (l < count) local = l;
    };
    // issues the kPerItem loads of an item into buffer B (nothing here waits)
    auto request = [&](TabBuffer& B, unsigned long long item) __attribute__((always_inline)) {
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        const bool real = local >= 0;
        const int lane = fresh_lane();
        // (the plan's length and row offset come with its tables: JointTable::len, start[kMaxSegments + 1]; holes re-read plan `first`)
        // tables: per joint slot one load of up to 57 word pairs (the packed form, landing at the end of the JointTable it is
        // expanded into); a descriptor over the two tiles the item's joints can lie in, one 32-bit offset per lane
        const unsigned long long l0 = (unsigned long long)(real ? local : 0) * dof + (real ? j0 : 0);       // wave-uniform
        const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned long long*>(tables) + (l0 >> 6) * (unsigned long long)(kPackedWords * 64), 0, (int)(2u * kTileBytes), 0x00020000);
        // where joint slot x starts inside the descriptor, for all slots at once in lanes 0..6 (as scalar code this is the
        // larger part of the loader's instructions, and the scalar unit is what a CU full of these blocks runs out of)
        const unsigned in_tile = (unsigned)(l0 & 63ull) + ((real && lane < nj) ? (unsigned)lane : 0u);
        const unsigned slot_base = (in_tile >> 6) * kTileBytes + (in_tile & 63u) * 16u;
        const unsigned lane_off = (unsigned)lane * 1024u;
        const bool wanted = lane < (whole_tables ? kPackedWords / 2 : kCappedPairs);
#pragma unroll
        for (int x = 0; x < kTabJointGroup; ++x) {
            const unsigned base = (unsigned)__builtin_amdgcn_readlane((int)slot_base, x);
            if (wanted)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_tab, (lds_ptr)(reinterpret_cast<char*>(&B.jt[x]) + kPackedByte), 16, base + lane_off, 0, 0, 0);
        }
    };
    // header of an item whose loads are in: what the streaming waves read
    auto publish_header = [&](TabBuffer& B, unsigned long long item) __attribute__((always_inline)) {
        const int lane = fresh_lane();
        if (item >= total) {
            if (lane == 0) lds_poke32(lds_offset(&B.hdr.done), 1);
            return;
        }
        long long local; int j0, nj;
        decode(item, local, j0, nj);
        int slen = 0;
        unsigned long long rel = 0ull;
        if (local >= 0) {
            // everything the header needs from what the loads brought, in one LDS round trip: lanes 0..6 read the joints' run
            // counts, lane 7 the trajectory length, lane 8 the plan's row offset (all written by the table pass; packed word 0 =
            // nseg | len, the row offset is the upper half of packed word 11)
            const unsigned pk0 = lds_offset(&B.jt[0]) + kPackedByte;
            const unsigned peek_at = lane < kTabJointGroup ? lds_offset(&B.jt[lane]) + kPackedByte
                                     : (lane == kTabJointGroup ? pk0 + 4u : pk0 + (unsigned)(1 + (kMaxSegments + 1) / 2) * 8u + 4u * ((kMaxSegments + 1) & 1));
            const int peeked = lds_peek32(peek_at);
            const int len = __builtin_amdgcn_readlane(peeked, kTabJointGroup);
            slen = stored_len(len, rows);
            rel = (unsigned long long)(unsigned)__builtin_amdgcn_readlane(peeked, kTabJointGroup + 1) * (unsigned long long)kRowAlign;
            const unsigned long long stride = ((unsigned long long)slen + (kRowAlign - 1)) / kRowAlign * kRowAlign;
            if (slen > 0 && rel + 4ull * dof * stride > capacity) {
                if (lane == 0 && j0 == 0) atomicOr(&rec.status[first + local], kStatusOverflow);
                slen = 0;
            }
            // the largest run count among the item's joints (wave-uniform): how many passes the expansion needs
            const int runs_of_mine = lane < nj ? peeked : 0;
            int max_runs = 0;
#pragma unroll
            for (int x = 0; x < kTabJointGroup; ++x) {
                const int n_x = __builtin_amdgcn_readlane(runs_of_mine, x);
                max_runs = n_x > max_runs ? n_x : max_runs;
            }
            if (slen > 0 && !whole_tables && max_runs > kCappedRuns) {
                // a capped row whose joint has more than 8 runs inside the cap (short trajectories): fetch the rest now
                if (stamps && lane == 0) stamps[8 * item + 2] = 1ull;
                const unsigned long long l0 = (unsigned long long)local * dof + j0;
                const __amdgpu_buffer_rsrc_t r_tab = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<unsigned long long*>(tables) + (l0 >> 6) * (unsigned long long)(kPackedWords * 64), 0, (int)(2u * kTileBytes), 0x00020000);
#pragma unroll
                for (int x = 0; x < kTabJointGroup; ++x) {
                    const unsigned long long li = l0 + (unsigned)(x < nj ? x : 0);
                    const unsigned base = (unsigned)((li >> 6) - (l0 >> 6)) * kTileBytes + (unsigned)(li & 63ull) * 16u;
                    if (lane >= kCappedPairs && lane < kPackedWords / 2)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(r_tab, (lds_ptr)(reinterpret_cast<char*>(&B.jt[x]) + kPackedByte), 16, base + (unsigned)lane * 1024u, 0, 0, 0);
                }
                LTP_WAIT_VMCNT(0);                          // rare: drain everything (later waits only get easier)
            } else if (stamps && lane == 0 && slen > 0 && !whole_tables) {
                stamps[8 * item + 2] = 0ull;
            }
            // (expanding in the streaming wave that owns the joint instead — seven waves in parallel — was measured: the sampler
            // of 64-sample rows went from 4.1 to 4.9 ms; the streaming waves are the longer side of an item already)
            if (slen > 0) expand_packed_tables(B, nj, max_runs, lane, t_sample);
        }
        if (lane == 0) {
            lds_poke64(lds_offset(&B.hdr.rel), rel);
            lds_poke32(lds_offset(&B.hdr.slen), slen);
            lds_poke32(lds_offset(&B.hdr.j0), j0);
            lds_poke32(lds_offset(&B.hdr.nj), nj);
            lds_poke32(lds_offset(&B.hdr.done), 0);
            lds_poke64(lds_offset(&B.hdr.item), item);
        }
    };
    // waits until every streaming wave is past the item that last used buffer seq % kTabBuffers (lane w < 7 watches wave w)
    auto wait_buffer_free = [&](int seq) __attribute__((always_inline)) {
        if (seq < kTabBuffers) return;
        const int need = seq - kTabBuffers + 1;
        const int lane = fresh_lane();
        const unsigned watch = lds_offset(&s_consumed[lane < kTabStreamWaves ? lane : 0]);
        for (;;) {
            const int c = lds_peek32(watch);
            if (!__builtin_amdgcn_ballot_w64(c < need)) break;
            __builtin_amdgcn_s_sleep(1);
        }
    };

    // items in flight: seq (to be published next) .. seq + kTabAhead - 1; their queue positions
    unsigned long long ids[kTabAhead + 1];
#pragma unroll
    for (int d = 0; d < kTabAhead; ++d) {
        ids[d] = next_item_id();
        request(buf[d], ids[d]);
        finish_draw();
    }
    unsigned long long t_prev_pub = 0ull;
    for (int seq = 0;; ++seq) {
        // request item seq + kTabAhead into its buffer, as soon as the streaming waves have left it (the spin loop comes
        // before the queue draw: between the draw's atomic and its use there must be no loop, see next_item_id)
        wait_buffer_free(seq + kTabAhead);
        const unsigned long long t_top = stamps ? wall_clock64() : 0ull;
        ids[kTabAhead] = next_item_id();
        request(buf[(seq + kTabAhead) % kTabBuffers], ids[kTabAhead]);
        const unsigned long long t_req = stamps ? wall_clock64() : 0ull;
        // item seq is in when at most the loads of the kTabAhead younger items are outstanding
        LTP_WAIT_VMCNT(kTabAhead * kTabJointGroup);
        TabBuffer& B = buf[seq % kTabBuffers];
        const unsigned long long item = ids[0];
        if (stamps && fresh_lane() == 0 && item < total) stamps[8 * item + 1] = wall_clock64();
        publish_header(B, item);
        // Publishing is a plain LDS write: the table data is in LDS (the wait above), the header was written by this same
        // lane and LDS serves a wave's requests in order.
        if (fresh_lane() == 0) lds_poke32(lds_offset(&s_ready[seq % kTabBuffers]), seq + 1);
        if (stamps && fresh_lane() == 0 && item < total) { stamps[8 * item + 3] = wall_clock64(); stamps[8 * item + 0] = t_top; stamps[8 * item + 6] = t_req; stamps[8 * item + 7] = t_prev_pub; }
        if (stamps) t_prev_pub = wall_clock64();
        finish_draw();
        if (item >= total) break;                       // the item just published says done: everyone leaves on reading it
#pragma unroll
        for (int d = 0; d < kTabAhead; ++d) ids[d] = ids[d + 1];
    }
    LTP_WAIT_VMCNT(0);                                  // nothing may still be landing in LDS when the wave ends
}

// The register budget decides how many streaming waves a CU holds, and attributes cannot depend on template parameters:
// one kernel per row type. float64: 3 blocks of 8 waves per CU (21 streaming waves, 6 waves per SIMD, <= 80 VGPRs);
// float32 (4 samples per lane in flight, 116 VGPRs; at 80 it spills inside the store loop): 2 blocks (14 streaming waves).
#define LTP_TAB_KERNEL(NAME, ST, TY, WAVES)                                                                                          \
    __global__ void __launch_bounds__(kTabThreads) __attribute__((amdgpu_waves_per_eu(WAVES, 8)))                                    \
    NAME(long long first, long long count, long long base_first, int dof, Records rec, const unsigned long long* __restrict__ offsets, \
         TY* __restrict__ out, unsigned long long capacity, int spread, RowSpec rows, unsigned long long* __restrict__ next_item,     \
         const unsigned long long* __restrict__ tables, int draw_chunk, unsigned long long* __restrict__ stamps, double t_sample)     \
    {                                                                                                                                 \
        sample_tab_body<ST, TY>(first, count, base_first, dof, rec, offsets, out, capacity, spread, rows, next_item, tables, draw_chunk, stamps, t_sample); \
    }
LTP_TAB_KERNEL(k_sample_tab_f64, false, double, 6)
LTP_TAB_KERNEL(k_sample_tab_f64_nt, true, double, 6)
LTP_TAB_KERNEL(k_sample_tab_f32, false, float, 4)
LTP_TAB_KERNEL(k_sample_tab_f32_nt, true, float, 4)
#undef LTP_TAB_KERNEL

int sample_tab_resident_blocks(int device, bool f32)
{
    int cus = 0, per_cu = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256;
    hipError_t e = f32 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample_tab_f32_nt, kTabThreads, 0)
                       : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_sample_tab_f64_nt, kTabThreads, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = f32 ? 2 : 3;
    return cus * per_cu;
}

void launch_sample_tab(hipStream_t s, long long first, long long count, long long base_first, int dof, Records rec,
                       const unsigned long long* offsets, void* out, bool f32, unsigned long long capacity, int flags, RowSpec rows,
                       unsigned long long* next_item, int resident_blocks, const unsigned long long* tables, double t_sample,
                       unsigned long long* stamps)
{
    if (count <= 0) return;
    int spread = (flags >> 8) & 0xFFFF;
    if (spread == 0) spread = kSampleSpread;
    const int ngroups = (dof + kTabJointGroup - 1) / kTabJointGroup;
    long long blocks = resident_blocks > 0 ? resident_blocks : 768;
    const dim3 block(kTabThreads);
    // (the loader pays one exposed atomic round trip per draw: larger chunks than k_sample's)
    const int draw_chunk = 2 * queue_draw_chunk(rows, f32, dof < kTabJointGroup ? dof : kTabJointGroup);
    if ((long long)spread > count) spread = (int)count;
    if (blocks > count * ngroups) blocks = count * ngroups;
    const dim3 grid((unsigned)blocks);
#define LTP_TAB_CASE(K, TY) hipLaunchKernelGGL(K, grid, block, 0, s, first, count, base_first, dof, rec, offsets, (TY*)out, capacity, spread, rows, next_item, tables, draw_chunk, stamps, t_sample)
    switch ((flags & 1) | (f32 ? 2 : 0)) {
    case 0: LTP_TAB_CASE(k_sample_tab_f64, double); break;
    case 1: LTP_TAB_CASE(k_sample_tab_f64_nt, double); break;
    case 2: LTP_TAB_CASE(k_sample_tab_f32, float); break;
    default: LTP_TAB_CASE(k_sample_tab_f32_nt, float); break;
    }
#undef LTP_TAB_CASE
}

}  // namespace ltp
