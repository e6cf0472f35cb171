// dk_bucket_seg.h -- per-segment set kernels: seg_insert / seg_probe (Bloom) and seg_exact_* (exact set), slice reductions
// (part of the bucketed kernel family: dk_kernels_bucket.h has the overview and includes the parts in order)
#pragma once
#include "dk_bucket_common.h"

namespace dk {

// ---- per-segment kernels ----------------------------------------------------------------------------
// A segment's records are the concatenation of n_pieces pieces:
//   piece r of segment s = recs[(s * n_pieces + r) * piece_cap ...], cnt[s * n_pieces + r] records
template <class R>
struct PieceList {
    const R *recs;
    const uint32_t *cnt;
    uint32_t n_pieces;     // <= MAX_R
    uint32_t piece_cap;
    // optional extra records per segment (overflow records that were absent), CSR over segments:
    // segment s owns extra[extra_off[s] .. extra_off[s + 1])
    const R *extra;
    const uint32_t *extra_off;
    // Sub-segment split (set kernels only): the list is partitioned into REGIONS of 2^sbits 64-KiB segments; the
    // workgroup of segment g reads the records of region g >> sbits and takes those whose hash bits at sub_shift
    // equal g's low sbits.  The last partition bits are thus resolved by 2^sbits workgroups re-reading one region
    // (mostly from L2 / Infinity Cache) instead of by one more multisplit pass over HBM.
    int sbits = 0, sub_shift = 0;
    // Packed regions (set kernels with PK): region g of the launch holds piece_cap packed records (dk_bucket_common.h) whose
    // hashes all start with the region's global index, pk_region0 + g, in their top pk_bits bits
    int pk_bits = 0;
    uint64_t pk_region0 = 0;
    // Piece-major layout (the receiving side of a multi-GPU exchange: n_pieces slices, one per rank, each holding the
    // same n_segs units): piece r of segment s = recs[(r * n_segs + s) * piece_cap ...], cnt[r * n_segs + s].  0 = segment-major.
    uint64_t n_segs = 0;
    __device__ __forceinline__ bool mine(uint64_t seg_id, uint64_t h) const
    {
        return sbits == 0 || ((uint32_t)(h >> sub_shift) & ((1u << sbits) - 1u)) == ((uint32_t)seg_id & ((1u << sbits) - 1u));
    }
};

template <class R>
struct SegPieces {
    uint32_t start[MAX_R + 1];     // prefix sums of the piece sizes; start[MAX_R] = records in the pieces
    const R *base;                 // first piece of the segment
    uint64_t piece_stride;         // records between two pieces of the segment
    const R *extra;                // extra records of the segment (or nullptr)
    uint32_t n_extra;
    bool single;                   // one piece (the usual case): no search for the piece of a record
    __device__ __forceinline__ uint32_t total() const { return start[MAX_R] + n_extra; }
    __device__ __forceinline__ R at(uint32_t i) const
    {
        if (i >= start[MAX_R]) return extra[i - start[MAX_R]];
        if (single) return base[i];
        uint32_t r = 0, st = 0;
#pragma unroll
        for (int q = 1; q < MAX_R; q++)
            if (i >= start[q]) { r = (uint32_t)q; st = start[q]; }     // starts are non-decreasing
        return base[(uint64_t)r * piece_stride + (i - st)];
    }
};

// the loads of seg_pieces, separable so that a persistent kernel can issue them one segment ahead
struct SegCounts {
    uint32_t c[MAX_R];
    uint32_t o0, o1;
};

template <class R>
__device__ __forceinline__ SegCounts seg_counts(const PieceList<R> &pl, uint64_t seg_id)
{
    SegCounts sc;
#pragma unroll
    for (int q = 0; q < MAX_R; q++)
        sc.c[q] = (uint32_t)q < pl.n_pieces ? pl.cnt[pl.n_segs ? (uint64_t)q * pl.n_segs + seg_id : seg_id * pl.n_pieces + q] : 0u;
    sc.o0 = sc.o1 = 0;
    if (pl.extra) {
        sc.o0 = pl.extra_off[seg_id];
        sc.o1 = pl.extra_off[seg_id + 1];
    }
    return sc;
}

template <class R>
__device__ __forceinline__ SegPieces<R> seg_pieces(const PieceList<R> &pl, uint64_t seg_id, const SegCounts &sc)
{
    SegPieces<R> sp;
    sp.base = pl.recs + (pl.n_segs ? seg_id : seg_id * pl.n_pieces) * (uint64_t)pl.piece_cap;
    sp.piece_stride = pl.n_segs ? pl.n_segs * (uint64_t)pl.piece_cap : (uint64_t)pl.piece_cap;
    uint32_t acc = 0;
#pragma unroll
    for (int q = 0; q < MAX_R; q++) {
        sp.start[q] = acc;
        acc += sc.c[q] < pl.piece_cap ? sc.c[q] : pl.piece_cap;
    }
    sp.start[MAX_R] = acc;
    sp.single = pl.n_pieces == 1;
    sp.extra = pl.extra ? pl.extra + sc.o0 : nullptr;
    sp.n_extra = sc.o1 - sc.o0;
    return sp;
}

template <class R>
__device__ __forceinline__ SegPieces<R> seg_pieces(const PieceList<R> &pl, uint64_t seg_id)
{
    return seg_pieces(pl, seg_id, seg_counts(pl, seg_id));
}

__device__ __forceinline__ void load_segment(uint32_t *seg, const unsigned long long *filter, uint64_t seg_id)
{
    const uint4 *src = (const uint4 *)filter + seg_id * (SEG_BYTES / 16);
    uint4 *dst = (uint4 *)seg;
    for (int i = (int)threadIdx.x; i < SEG_BYTES / 16; i += (int)blockDim.x) dst[i] = src[i];
}

// Which 64-KiB segment a workgroup of the set kernels takes: simply its block index.  (With a sub-segment split,
// placing the 2^sbits workgroups that share a region on one XCD -- block indices 8 apart, so that the region's second
// reading could hit that XCD's L2 -- changed nothing: seg_probe 9.32 vs 9.28 ms at 2^39 bits, two hash windows, 24 M
// reads.  Keeping the region's records in registers while one workgroup stages both segments in turn does not fit the
// 64 VGPRs that two workgroups per CU allow.)
__device__ __forceinline__ uint64_t segment_of_block() { return blockIdx.x; }

// The three dependent fetches of a segment workgroup -- piece sizes, the 64-KiB segment, the first records -- are
// issued back to back: the segment travels to registers while the sizes arrive, the first records are requested as
// soon as the sizes are known, and only then is the segment written to LDS (with two workgroups per CU every
// exposed round trip to HBM is a third of a workgroup's life).
constexpr int SEG_VEC = SEG_BYTES / 16 / SEG_THREADS;      // uint4 per thread per segment
struct SegRegs { uint4 v[SEG_VEC]; };

__device__ __forceinline__ SegRegs fetch_segment(const unsigned long long *filter, uint64_t seg_id)
{
    const uint4 *src = (const uint4 *)filter + seg_id * (SEG_BYTES / 16);
    SegRegs r;
#pragma unroll
    for (int q = 0; q < SEG_VEC; q++) {
        // (non-temporal: a segment is read once per sweep of the set; worth ~1 % of the set kernels)
        const uint4 *p = &src[q * SEG_THREADS + (int)threadIdx.x];
        r.v[q].x = __builtin_nontemporal_load(&p->x);
        r.v[q].y = __builtin_nontemporal_load(&p->y);
        r.v[q].z = __builtin_nontemporal_load(&p->z);
        r.v[q].w = __builtin_nontemporal_load(&p->w);
    }
    return r;
}

__device__ __forceinline__ void stage_segment(uint32_t *seg, const SegRegs &r)
{
    uint4 *dst = (uint4 *)seg;
#pragma unroll
    for (int q = 0; q < SEG_VEC; q++) dst[q * SEG_THREADS + (int)threadIdx.x] = r.v[q];
}

// seg_base: the launch covers the segments seg_base .. seg_base + gridDim.x of the set (one slab of a slab-wise partition;
// the piece list is indexed from the launch's first region)
// A segment goes back to HBM with non-temporal stores: it is not read again before the next sweep of the set, and
// keeping 64 KiB per workgroup out of the L2's way is worth 11 % of seg_insert (parent batch of 128 M reads: 51.6 -> 45.8 ms).
// (The same hint on the partition kernels' stores is a disaster -- scan_part 21.4 -> 29.6 ms, repart 20.0 -> 52.5: their short
// runs rely on the L2 to assemble whole lines -- and on seg_count's outputs it changes nothing.)
__device__ __forceinline__ void write_back_segment(uint4 *dst, const uint4 *s4)
{
    for (int i = (int)threadIdx.x; i < SEG_BYTES / 16; i += SEG_THREADS) {
        const uint4 v = s4[i];
        __builtin_nontemporal_store(v.x, &dst[i].x);
        __builtin_nontemporal_store(v.y, &dst[i].y);
        __builtin_nontemporal_store(v.z, &dst[i].z);
        __builtin_nontemporal_store(v.w, &dst[i].w);
    }
}

template <class R, bool PK = false>
DK_SEG_KERNEL
seg_insert_kernel(unsigned long long *filter, PieceList<R> pl, int n_hashes, int blk_shift, uint64_t seg_base)
{
    __shared__ __attribute__((aligned(16))) uint32_t seg[SEG_WORDS32];
    const uint64_t seg_id = segment_of_block();
    const SegCounts sc = seg_counts(pl, seg_id >> pl.sbits);
    const SegRegs sr = fetch_segment(filter, seg_base + seg_id);
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) return;                       // nothing to add: leave the segment untouched
    constexpr int UNROLL = 8;
    uint64_t h[UNROLL];
    bool have[UNROLL];
    const uint64_t region = seg_id >> pl.sbits, prefix = PK ? (pl.pk_region0 + region) << (64 - pl.pk_bits) : 0;
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            if constexpr (PK) h[u] = packed_load(pl.recs, region, pl.piece_cap, have[u] ? i : 0, prefix);
            else h[u] = sp.at(have[u] ? i : 0).h;
        }
    };
    fetch(0);
    stage_segment(seg, sr);
    __syncthreads();
    for (uint32_t i0 = 0;;) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (!have[u] || !pl.mine(seg_id, h[u])) continue;
            const uint32_t blk = (uint32_t)(h[u] >> blk_shift) & (SEG_BLOCKS - 1);
            const uint32_t a = (uint32_t)(h[u] & 511), d = (uint32_t)((h[u] >> 9) & 511) | 1u;
            for (int j = 0; j < n_hashes; j++) {
                const uint32_t bit = (a + (uint32_t)j * d) & 511;
                atomicOr(&seg[blk * 16 + (bit >> 5)], 1u << (bit & 31));
            }
        }
        i0 += UNROLL * SEG_THREADS;
        if (i0 >= n) break;
        fetch(i0);
    }
    __syncthreads();
    uint4 *dst = (uint4 *)filter + (seg_base + seg_id) * (SEG_BYTES / 16);
    const uint4 *s4 = (const uint4 *)seg;
    write_back_segment(dst, s4);
}

// Where the absent records of a segment go.
//   per batch (ACC = 0): recs[seg * cap ...], compacted per wave by ballot; cnt[seg] = their number
//   accumulate (ACC = 1 / 2, dk_accum_add): the accumulator's counting units of the segment -- unit = seg << sub_bits |
//     the next sub_bits hash bits -- appended behind cnt[unit], which persists from batch to batch; a record whose unit
//     is full goes to the accumulator's overflow list.  ACC = 2: the units hold PACKED records (below).
// seg = the segment's index inside the launch (= blockIdx.x); the filter is addressed with seg_base + seg.
template <class R>
struct MissOut {
    R *recs;
    uint32_t cap;
    uint32_t *cnt;
    int sub_bits, sub_shift;
    OvfList<R> ovf;
};
constexpr int MAX_SUB_BITS = 10;
constexpr int SUB_TALLY = 1 << MAX_SUB_BITS;       // index of the batch's absent tally behind the units' fills
constexpr int ACC_NONE = 0, ACC_PLAIN = 1, ACC_PACKED = 2;

template <class R, int ACC>
struct MissSink {
    uint32_t *sfill;       // LDS: ACC: fill of the segment's units; else [0] = absent records so far
    R *dst;
    const MissOut<R> &mo;
    uint64_t seg;
    uint32_t n_dropped = 0;
    __device__ __forceinline__ MissSink(uint32_t *lds, const MissOut<R> &m, uint64_t seg_local) : sfill(lds), mo(m), seg(seg_local)
    {
        if constexpr (ACC != ACC_NONE) {
            dst = ACC == ACC_PACKED ? m.recs : m.recs + (seg_local << m.sub_bits) * (uint64_t)m.cap;
            if (threadIdx.x < (1u << m.sub_bits)) sfill[threadIdx.x] = m.cnt[(seg_local << m.sub_bits) + threadIdx.x];
            if (threadIdx.x == 0) sfill[SUB_TALLY] = 0;                    // absent records of this batch
        } else {
            dst = m.recs + seg_local * (uint64_t)m.cap;
            if (threadIdx.x == 0) sfill[0] = 0;
        }
    }
    // every lane of the wave calls this (ballots inside)
    __device__ __forceinline__ void put(bool absent, const R &rec)
    {
        if constexpr (ACC != ACC_NONE) {
            uint32_t sub = 0, pos = 0;
            if (absent) {
                sub = (uint32_t)(rec.h >> mo.sub_shift) & ((1u << mo.sub_bits) - 1u);
                pos = atomicAdd(&sfill[sub], 1u);
            }
            const bool full = absent && pos >= mo.cap;
            if (absent && !full) {
                if constexpr (ACC == ACC_PACKED) packed_store(dst, (seg << mo.sub_bits) + sub, mo.cap, pos, rec.h);
                else dst[(uint64_t)sub * mo.cap + pos] = rec;
            }
            if (__ballot(full)) ovf_append(mo.ovf, full, rec, n_dropped);
        } else {
            const uint64_t b = __ballot(absent);
            if (b) {
                const int leader = __ffsll((long long)b) - 1;
                uint32_t wbase = 0;
                if (lane_id() == leader) wbase = atomicAdd(&sfill[0], (uint32_t)__popcll(b));
                wbase = __shfl(wbase, leader);
                if (absent) dst[wbase + popc_below(b)] = rec;
            }
        }
    }
    // after a workgroup barrier; my_absent = absent records this thread saw (ACC only)
    __device__ __forceinline__ void finish(Counters *ctr, uint32_t my_absent)
    {
        if constexpr (ACC != ACC_NONE) {
            const uint32_t ws = wave_total(my_absent);
            if (lane_id() == 0 && ws) atomicAdd(&sfill[SUB_TALLY], ws);
            lds_barrier();
            if (threadIdx.x < (1u << mo.sub_bits)) {
                const uint32_t f = sfill[threadIdx.x];
                mo.cnt[(seg << mo.sub_bits) + threadIdx.x] = f < mo.cap ? f : mo.cap;
            }
            if (threadIdx.x == 0 && sfill[SUB_TALLY]) atomicAdd(&ctr->shard[blockIdx.x % COUNTER_SHARDS], (unsigned long long)sfill[SUB_TALLY]);
            n_dropped = (uint32_t)wave_sum(n_dropped);
            if (lane_id() == 0 && n_dropped) atomicAdd(&ctr->n_sink_drop, (unsigned long long)n_dropped);
        } else {
            if (threadIdx.x == 0) {
                mo.cnt[seg] = sfill[0];
                if (sfill[0]) atomicAdd(&ctr->shard[blockIdx.x % COUNTER_SHARDS], (unsigned long long)sfill[0]);
            }
        }
    }
};

// An accumulating launch gives up before it appends anything when an earlier kernel of the batch has lost records --
// the partition of this slab or of an earlier one (Counters::n_overflow), or the sink of an earlier slab
// (n_sink_drop): the host then knows exactly which slabs reached the accumulator (Counters::fail_mark) and redoes the rest.
__device__ __forceinline__ bool batch_has_failed(const Counters *ctr)
{
    return (ctr->n_overflow | ctr->n_sink_drop) != 0;
}

// NH > 0: the number of hash bits is a compile-time constant (the four LDS reads of a record are then
// issued back to back instead of one by one behind the short-circuit test); NH == 0: n_hashes at run time
template <class R, int NH, int ACC, bool PK = false>
DK_SEG_KERNEL
seg_probe_kernel(const unsigned long long *__restrict__ filter, PieceList<R> pl, int n_hashes, int blk_shift,
                 uint64_t seg_base, MissOut<R> mo, Counters *ctr)
{
    __shared__ __attribute__((aligned(16))) uint32_t seg[SEG_WORDS32];
    __shared__ uint32_t sfill[ACC ? SUB_TALLY + 1 : 1];
    const uint64_t seg_id = segment_of_block();
    const bool no_set = filter == nullptr;    // accumulating KmerCounter: every record counts as absent
    const SegCounts sc = seg_counts(pl, seg_id >> pl.sbits);
    SegRegs sr;
    if (!no_set) sr = fetch_segment(filter, seg_base + seg_id);
    if (ACC != ACC_NONE && batch_has_failed(ctr)) return;
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) {
        if (!ACC && threadIdx.x == 0) mo.cnt[seg_id] = 0;
        return;
    }
    constexpr int UNROLL = 8;                 // records in flight per thread: loads first, then the LDS tests
    R rec[UNROLL];
    bool have[UNROLL];
    // A wave whose slot u lies beyond the segment's records skips the slot altogether (and every later one): with 3.7 K
    // records per segment -- a whole-genome batch against 2^19 segments -- more than half of the 8192 slots of an iteration
    // are empty, and the tests of an empty slot cost what those of a record cost.
    const uint32_t wave_first = (uint32_t)__builtin_amdgcn_readfirstlane((int)threadIdx.x);
    const uint64_t region = seg_id >> pl.sbits, prefix = PK ? (pl.pk_region0 + region) << (64 - pl.pk_bits) : 0;
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            have[u] = false;
            if (i0 + (uint32_t)u * SEG_THREADS + wave_first >= n) break;
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            if constexpr (PK) rec[u].h = packed_load(pl.recs, region, pl.piece_cap, have[u] ? i : 0, prefix);
            else rec[u] = sp.at(have[u] ? i : 0);
        }
    };
    fetch(0);
    MissSink<R, ACC> sink(sfill, mo, seg_id);
    if (!no_set) stage_segment(seg, sr);
    __syncthreads();
    uint32_t my_absent = 0;
    for (uint32_t i0 = 0;;) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            if (i0 + (uint32_t)u * SEG_THREADS + wave_first >= n) break;
            const uint64_t hu = rec[u].h;
            const uint32_t blk = (uint32_t)(hu >> blk_shift) & (SEG_BLOCKS - 1);
            const uint32_t a = (uint32_t)(hu & 511), d = (uint32_t)((hu >> 9) & 511) | 1u;
            bool all = true;
            if constexpr (NH > 0) {
                uint32_t acc = 1u;
#pragma unroll
                for (int j = 0; j < NH; j++) {
                    const uint32_t bit = (a + (uint32_t)j * d) & 511;
                    acc &= seg[blk * 16 + (bit >> 5)] >> (bit & 31);
                }
                all = acc & 1u;
            } else {
                all = !no_set;
                for (int j = 0; j < n_hashes; j++) {
                    const uint32_t bit = (a + (uint32_t)j * d) & 511;
                    all = all && ((seg[blk * 16 + (bit >> 5)] >> (bit & 31)) & 1u);
                }
            }
            const bool absent = have[u] && !all && pl.mine(seg_id, hu);
            my_absent += absent;
            sink.put(absent, rec[u]);
        }
        i0 += UNROLL * SEG_THREADS;
        if (i0 >= n) break;
        fetch(i0);
    }
    __syncthreads();
    sink.finish(ctr, my_absent);
}

// ---- exact set: the segment is an open-addressing table (dk_device.h) -----------------------------
// Same shape as seg_insert / seg_probe: segment -> LDS, one LDS operation chain per record, segment back.
template <class R, bool PK = false>
DK_SEG_KERNEL
seg_exact_insert_kernel(unsigned long long *table, PieceList<R> pl, int T, Counters *ctr, uint64_t seg_base)
{
    constexpr bool WIDE = sizeof(R) == 16;
    static_assert(SEG_BYTES == EXACT_SEG_WORDS * 8, "exact segments are the filter segments");
    __shared__ __attribute__((aligned(16))) unsigned long long tab[EXACT_SEG_WORDS];
    const uint64_t seg_id = segment_of_block();
    const SegCounts sc = seg_counts(pl, seg_id >> pl.sbits);
    const SegRegs sr = fetch_segment(table, seg_base + seg_id);
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) return;
    constexpr int UNROLL = 8;
    R rec[UNROLL];
    bool have[UNROLL];
    const uint64_t region = seg_id >> pl.sbits, prefix = PK ? (pl.pk_region0 + region) << (64 - pl.pk_bits) : 0;
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            if constexpr (PK) rec[u].h = packed_load(pl.recs, region, pl.piece_cap, have[u] ? i : 0, prefix);
            else rec[u] = sp.at(have[u] ? i : 0);
        }
    };
    fetch(0);
    stage_segment((uint32_t *)tab, sr);
    __syncthreads();
    const uint64_t EMPTY = exact_empty(seg_base + seg_id, T);
    uint32_t n_full = 0;
    for (uint32_t i0 = 0;;) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++)
            if (have[u] && pl.mine(seg_id, rec[u].h) &&
                exact_insert<WIDE, __HIP_MEMORY_SCOPE_WORKGROUP>(tab, EMPTY, rec[u].h, rec_hi(rec[u])) == 2) n_full++;
        i0 += UNROLL * SEG_THREADS;
        if (i0 >= n) break;
        fetch(i0);
    }
    __syncthreads();
    uint4 *dst = (uint4 *)table + (seg_base + seg_id) * (SEG_BYTES / 16);
    const uint4 *s4 = (const uint4 *)tab;
    write_back_segment(dst, s4);
    n_full = (uint32_t)wave_sum(n_full);
    if (lane_id() == 0 && n_full) atomicAdd(&ctr->n_set_full, (unsigned long long)n_full);
}

// One workgroup per segment, eight records in flight per thread, 32-byte bucket probes.  (A persistent
// walk over the segments with the table and the records fetched in one round trip measured 6.8 ms
// against 5.1 ms for this form at 2^17 segments: the hardware's workgroup scheduler overlaps the
// segments' load / probe phases better than two resident persistent workgroups per CU do.)
template <class R, int ACC, bool PK = false>
DK_SEG_KERNEL
seg_exact_probe_kernel(const unsigned long long *__restrict__ table, PieceList<R> pl, int T, uint64_t seg_base,
                       MissOut<R> mo, Counters *ctr)
{
    constexpr bool WIDE = sizeof(R) == 16;
    __shared__ __attribute__((aligned(16))) unsigned long long tab[EXACT_SEG_WORDS];
    __shared__ uint32_t sfill[ACC ? SUB_TALLY + 1 : 1];
    const uint64_t seg_id = segment_of_block();
    const SegCounts sc = seg_counts(pl, seg_id >> pl.sbits);
    const SegRegs sr = fetch_segment(table, seg_base + seg_id);
    if (ACC != ACC_NONE && batch_has_failed(ctr)) return;
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) {
        if (!ACC && threadIdx.x == 0) mo.cnt[seg_id] = 0;
        return;
    }
    constexpr int UNROLL = 8;
    R rec[UNROLL];
    bool have[UNROLL];
    const uint64_t region = seg_id >> pl.sbits, prefix = PK ? (pl.pk_region0 + region) << (64 - pl.pk_bits) : 0;
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            if constexpr (PK) rec[u].h = packed_load(pl.recs, region, pl.piece_cap, have[u] ? i : 0, prefix);
            else rec[u] = sp.at(have[u] ? i : 0);
        }
    };
    fetch(0);
    MissSink<R, ACC> sink(sfill, mo, seg_id);
    stage_segment((uint32_t *)tab, sr);
    __syncthreads();
    const uint64_t EMPTY = exact_empty(seg_base + seg_id, T);
    uint32_t my_absent = 0;
    for (uint32_t i0 = 0;;) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const bool absent = have[u] && pl.mine(seg_id, rec[u].h) && !exact_find<WIDE>(tab, EMPTY, rec[u].h, rec_hi(rec[u]));
            my_absent += absent;
            sink.put(absent, rec[u]);
        }
        i0 += UNROLL * SEG_THREADS;
        if (i0 >= n) break;
        fetch(i0);
    }
    __syncthreads();
    sink.finish(ctr, my_absent);
}

// Union of table slices, the exact-set counterpart of or_slices_kernel: segment (first_seg + blockIdx.x)
// of dst is staged in LDS and every key of the same segment in each of the n_slices source slices is
// inserted into it (slot positions differ between tables built in different orders, so OR-ing is not an option).
template <bool WIDE>
__global__ void __launch_bounds__(SEG_THREADS)
union_slices_kernel(unsigned long long *dst, const unsigned long long *__restrict__ src, uint64_t n_slices,
                    uint64_t slice_words, uint64_t first_seg, int T, Counters *ctr)
{
    __shared__ __attribute__((aligned(16))) unsigned long long tab[EXACT_SEG_WORDS];
    const uint64_t seg_local = blockIdx.x;
    load_segment((uint32_t *)tab, dst, seg_local);
    __syncthreads();
    const uint64_t EMPTY = exact_empty(first_seg + seg_local, T);
    uint32_t n_full = 0;
    for (uint64_t j = 0; j < n_slices; j++) {
        const unsigned long long *sseg = src + j * slice_words + seg_local * EXACT_SEG_WORDS;
        for (uint32_t i = threadIdx.x; i < ExactGeom<WIDE>::SLOTS; i += SEG_THREADS) {
            uint64_t h, hi = 0;
            if constexpr (WIDE) {
                const ulonglong2 v = *(const ulonglong2 *)(sseg + 2 * i);
                h = v.x;
                hi = v.y;
            } else {
                h = sseg[i];
            }
            if (h != EMPTY && exact_insert<WIDE, __HIP_MEMORY_SCOPE_WORKGROUP>(tab, EMPTY, h, hi) == 2) n_full++;
        }
    }
    __syncthreads();
    uint4 *out = (uint4 *)dst + seg_local * (SEG_BYTES / 16);
    const uint4 *s4 = (const uint4 *)tab;
    for (int i = (int)threadIdx.x; i < SEG_BYTES / 16; i += SEG_THREADS) out[i] = s4[i];
    n_full = (uint32_t)wave_sum(n_full);
    if (lane_id() == 0 && n_full) atomicAdd(&ctr->n_set_full, (unsigned long long)n_full);
}

}  // namespace dk
