// dk_bucket_rare.h -- rare paths: overflow records one by one, accumulator appends through global cursors
// (part of the bucketed kernel family: dk_kernels_bucket.h has the overview and includes the parts in order)
#pragma once
#include "dk_bucket_seg.h"

namespace dk {

// ---- overflow records: exact one-by-one handling (rare path) -------------------------------------------
__device__ __forceinline__ bool ovf_filter_op(unsigned long long *filter, uint64_t h, int log2_blocks, int n_hashes, bool set)
{
    unsigned long long *blk = filter + bloom_block(h, log2_blocks) * 8;
    const uint32_t a = (uint32_t)(h & 511), d = (uint32_t)((h >> 9) & 511) | 1u;
    bool all = true;
    for (int j = 0; j < n_hashes; j++) {
        const uint32_t bit = (a + (uint32_t)j * d) & 511;
        const unsigned long long m = 1ULL << (bit & 63);
        if (set) { if (!(blk[bit >> 6] & m)) atomicOr(&blk[bit >> 6], m); }
        else all = all && (blk[bit >> 6] & m);
    }
    return all;
}

// OR the overflow records into the filter (after seg_insert has written its segments back)
template <class R>
__global__ void __launch_bounds__(DIRECT_BLOCK)
ovf_insert_kernel(unsigned long long *filter, OvfList<R> ovf, int log2_blocks, int n_hashes, int exact_T, Counters *ctr)
{
    unsigned long long n = *ovf.count;
    if (n > ovf.cap) n = ovf.cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t n_full = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const R rec = ovf.recs[i];
        if (exact_T) {
            if (exact_insert_global<sizeof(R) == 16>(filter, exact_T, rec.h, rec_hi(rec)) == 2) n_full++;
        } else {
            ovf_filter_op(filter, rec.h, log2_blocks, n_hashes, true);
        }
    }
    if (n_full) atomicAdd(&ctr->n_set_full, (unsigned long long)n_full);
}

// Probe the overflow records (filter == nullptr: every record counts as absent); absent ones are
// appended to `miss` and tallied per segment for the CSR build.  h_lim != 0: only the records with h < h_lim (the slabs
// of a slab-wise accumulate that completed before the partition lost records).
template <class R>
__global__ void __launch_bounds__(DIRECT_BLOCK)
ovf_probe_kernel(unsigned long long *filter, OvfList<R> ovf, int log2_blocks, int n_hashes, int exact_T, int T,
                 uint64_t unit_base, R *__restrict__ miss, uint32_t *seg_hist, Counters *ctr, uint64_t h_lim = 0)
{
    unsigned long long n = *ovf.count;
    if (n > ovf.cap) n = ovf.cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n + 63) & ~63ULL;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        bool have = i < n;
        const R rec = ovf.recs[have ? i : 0];
        if (h_lim && rec.h >= h_lim) have = false;
        bool absent = have;
        if (have && filter)
            absent = exact_T ? !exact_contains<sizeof(R) == 16>(filter, exact_T, rec.h, rec_hi(rec))
                             : !ovf_filter_op(filter, rec.h, log2_blocks, n_hashes, false);
        const uint64_t slot = wave_append(absent, &ctr->n_ovf_miss);
        if (absent) {
            miss[slot] = rec;
            if (seg_hist) atomicAdd(&seg_hist[(rec.h >> (64 - T)) - unit_base], 1u);
        }
    }
}

// exclusive scan of seg_hist[n] into off[n + 1] by one workgroup (n <= 2^21)
__global__ void __launch_bounds__(1024)
ovf_scan_kernel(const uint32_t *__restrict__ hist, uint32_t *__restrict__ off, uint32_t n)
{
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t total;
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = lo + per < n ? lo + per : n;
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; i++) sum += hist[i];
    uint32_t run = block_excl_scan(sum, wave_sums, &total);
    for (uint32_t i = lo; i < hi; i++) { off[i] = run; run += hist[i]; }
    if (threadIdx.x == 0) off[n] = total;
}

// place the absent overflow records into their unit's slice (unit = top T hash bits - unit_base)
template <class R>
__global__ void __launch_bounds__(DIRECT_BLOCK)
ovf_scatter_kernel(const R *__restrict__ miss, uint64_t n, int T, uint64_t unit_base, const uint32_t *__restrict__ off,
                   uint32_t *fill, R *__restrict__ extra, uint64_t n_units = ~0ULL)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const R rec = miss[i];
        const uint64_t seg = (rec.h >> (64 - T)) - unit_base;
        if (seg >= n_units) continue;                   // another rank's hash range
        extra[off[seg] + atomicAdd(&fill[seg], 1u)] = rec;
    }
}

// ---- accumulator (dk_accum_*): rare-path appends through global cursors -------------------------------------
// histogram of a record list over the accumulator's units (CSR build of its overflow list at finish)
template <class R>
__global__ void __launch_bounds__(DIRECT_BLOCK)
unit_hist_kernel(const R *__restrict__ recs, uint64_t n, int T, uint64_t unit_base, uint32_t *hist, uint64_t n_units)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t u = (recs[i].h >> (64 - T)) - unit_base;
        if (u < n_units) atomicAdd(&hist[u], 1u);       // else: another rank's hash range
    }
}

// sum over units of min(fill, cap): the records a piece list holds
__global__ void __launch_bounds__(DIRECT_BLOCK)
fill_sum_kernel(const uint32_t *__restrict__ fill, uint64_t n, uint32_t cap, unsigned long long *out)
{
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) acc += fill[i] < cap ? fill[i] : cap;
    acc = wave_sum(acc);
    if (lane_id() == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}

// append records (all of them absent, all inside the window) to their units: the absent overflow records of a batch
template <class R, bool PACKED>
__global__ void __launch_bounds__(DIRECT_BLOCK)
acc_append_kernel(const R *__restrict__ recs, uint64_t n, int T, uint64_t unit_base, MissOut<R> mo, Counters *ctr)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n + 63) & ~63ULL;
    uint32_t n_dropped = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool have = i < n;
        const R rec = recs[have ? i : 0];
        bool full = false;
        if (have) {
            const uint64_t unit = (rec.h >> (64 - T)) - unit_base;
            const uint32_t pos = atomicAdd(&mo.cnt[unit], 1u);       // may run past cap: readers clamp
            full = pos >= mo.cap;
            if (!full) {
                if constexpr (PACKED) packed_store(mo.recs, unit, mo.cap, pos, rec.h);
                else mo.recs[unit * mo.cap + pos] = rec;
            }
        }
        if (__ballot(full)) ovf_append(mo.ovf, full, rec, n_dropped);
    }
    n_dropped = (uint32_t)wave_sum(n_dropped);
    if (lane_id() == 0 && n_dropped) atomicAdd(&ctr->n_sink_drop, (unsigned long long)n_dropped);
}

// the same for k-mers (the candidate list of the direct family: the exact redo path of a batch whose partition
// overflowed); k-mers outside the window -- or below h_from, the part of the hash range that a slab-wise accumulate
// had already completed -- are skipped; the appended ones are tallied in Counters::shard
template <bool WIDE, bool PACKED>
__global__ void __launch_bounds__(DIRECT_BLOCK)
acc_append_kmers_kernel(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi, uint64_t n, uint64_t seed,
                        int wbits, uint32_t widx, int T, uint64_t unit_base, MissOut<typename RecOf<WIDE>::type> mo, Counters *ctr,
                        uint64_t h_from)
{
    using R = typename RecOf<WIDE>::type;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n + 63) & ~63ULL;
    uint32_t n_dropped = 0;
    uint64_t n_in = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        bool have = i < n;
        R rec;
        const uint64_t khi = (WIDE && have) ? hi[i] : 0;
        rec.h = fmix64((have ? lo[i] : 0) ^ hash_tweak<WIDE>(khi, seed));
        if constexpr (WIDE) rec.hi = khi;
        if (wbits && (uint32_t)(rec.h >> (64 - wbits)) != widx) have = false;
        if (rec.h < h_from) have = false;
        bool full = false;
        if (have) {
            n_in++;
            const uint64_t unit = (rec.h >> (64 - T)) - unit_base;
            const uint32_t pos = atomicAdd(&mo.cnt[unit], 1u);
            full = pos >= mo.cap;
            if (!full) {
                if constexpr (PACKED) packed_store(mo.recs, unit, mo.cap, pos, rec.h);
                else mo.recs[unit * mo.cap + pos] = rec;
            }
        }
        if (__ballot(full)) ovf_append(mo.ovf, full, rec, n_dropped);
    }
    n_dropped = (uint32_t)wave_sum(n_dropped);
    n_in = wave_sum(n_in);
    if (lane_id() == 0) {
        if (n_dropped) atomicAdd(&ctr->n_sink_drop, (unsigned long long)n_dropped);
        if (n_in) atomicAdd(&ctr->shard[blockIdx.x % COUNTER_SHARDS], (unsigned long long)n_in);
    }
}

}  // namespace dk
