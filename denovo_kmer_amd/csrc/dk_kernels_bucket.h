// dk_kernels_bucket.h -- "bucketed" kernel family (k <= 32): every random access of the hot path
// is moved from HBM into LDS.
//
//   scan_part   packed stream -> canonical k-mer -> hash h (a bijection of the k-mer, so h IS the
//               record); LDS multisplit of a tile of 8192 positions by the top b1 hash bits,
//               runs written coalesced to per-bin regions in HBM
//   repart      second multisplit level by the next b2 bits (filters above 2^9 segments)
//   seg_insert  one workgroup per 64-KiB filter segment: segment -> LDS, ds_or per record, back
//   seg_probe   segment -> LDS, test per record, absent records compacted per segment
//   seg_count   absent records of a segment -> LDS hash table -> (k-mer, count) appended
//
// The filter bits produced are identical to the direct family's (same hash, same geometry), so
// both families are checked against the same oracle.  HBM traffic per k-mer:
// 3L/(8(L-k+1)) + 8 (scan_part) + 16 (repart) + 8 + filter/batch (seg_*) instead of one random
// 64-B block per k-mer.
#pragma once
#include <math.h>
#include <stdlib.h>

#include "dk_internal.h"

namespace dk {

constexpr int SEG_LOG2_BLOCKS = 10;                    // 2^10 blocks of 64 B = 64 KiB per segment
constexpr int SEG_BLOCKS = 1 << SEG_LOG2_BLOCKS;
constexpr int SEG_WORDS32 = SEG_BLOCKS * 16;
constexpr int SEG_BYTES = SEG_BLOCKS * 64;

constexpr int PART_THREADS = 1024;
constexpr int PART_PER_THREAD = 8;
constexpr int PART_TILE = PART_THREADS * PART_PER_THREAD;   // positions (or records) per tile
constexpr int MAX_BIN_BITS = 9;
constexpr int MAX_BINS = 1 << MAX_BIN_BITS;
constexpr int CURSOR_STRIDE = 32;                      // level-1 cursors on separate 128-B lines

constexpr int SEG_THREADS = 1024;
constexpr int CNT_THREADS = 256;
constexpr int CNT_SLOTS = 4096;                        // LDS hash table of seg_count
constexpr uint32_t NO_RANK = 0xFFFFFFFFu;

struct BucketPlan {
    int T;                 // log2(number of segments)
    int b1, b2;            // bits split at level 1 / level 2 (b2 == 0: single level)
    uint32_t p1, p2;       // bins at each level
    uint64_t n_seg;
    uint32_t cap1, cap2;   // records per level-1 bin / per segment
    uint64_t n_max;        // upper bound on records of the batch
};

// exclusive prefix sum over the block; every thread calls it; *total gets the block sum
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wave_sums, uint32_t *total)
{
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int n_waves = (int)(blockDim.x >> 6);
    uint32_t inc = v;
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wave_sums[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        uint32_t w = lane < n_waves ? wave_sums[lane] : 0;
        uint32_t wi = w;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(wi, o);
            if (lane >= o) wi += t;
        }
        if (lane < n_waves) wave_sums[lane] = wi - w;       // exclusive wave offsets
        if (lane == n_waves - 1) *total = wi;
    }
    __syncthreads();
    return wave_sums[wave] + inc - v;
}

// ---- shared multisplit tail ------------------------------------------------------------------------
// A tile's records sit in registers (hs = hash, rk = rank inside its bin from the LDS count).
// Wave 0 turns the per-bin counts into tile offsets and reserves the global ranges; the global
// atomics stay in flight while every wave scatters its records into the LDS stage, and only the
// copy-out needs their result.  Three barriers per tile (A: counts done - by the caller,
// B: offsets ready, C: stage + global bases ready); the next tile's count phase needs no barrier
// because it touches only cnt[], which wave 0 re-zeroes before B.
template <int THREADS, int PER_THREAD>
struct SplitLds {
    uint64_t stage[THREADS * PER_THREAD];
    uint32_t cnt[MAX_BINS];     // must be zero on entry to the first tile
    uint32_t off[MAX_BINS];
    uint32_t delta[MAX_BINS];   // global index in bin = stage index + delta[bin]  (mod 2^32)
    uint32_t total;
};

constexpr int MAX_BINS_PER_LANE = MAX_BINS / 64;

template <int THREADS, int PER_THREAD, class BinOf>
__device__ __forceinline__ void multisplit_flush(SplitLds<THREADS, PER_THREAD> &L, const uint64_t (&hs)[PER_THREAD],
                                                 const uint32_t (&rk)[PER_THREAD], int nbins, BinOf bin_of,
                                                 uint32_t *cursor, int cursor_stride, uint64_t bin_base,
                                                 uint32_t cap, uint64_t *__restrict__ out,
                                                 uint32_t &n_records, uint32_t &n_overflow)
{
    const int tid = (int)threadIdx.x;
    const bool w0 = tid < 64;
    const int m = nbins > 64 ? nbins / 64 : 1;          // bins per lane of wave 0
    const int first = tid * m;
    uint32_t g[MAX_BINS_PER_LANE];                      // the only state wave 0 carries across barrier B
    if (w0) {
        uint32_t sum = 0;
#pragma unroll
        for (int q = 0; q < MAX_BINS_PER_LANE; q++) {
            if (q < m && first + q < nbins) { L.off[first + q] = sum; sum += L.cnt[first + q]; }
        }
        uint32_t inc = sum;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t t = __shfl_up(inc, o);
            if (tid >= o) inc += t;
        }
        const uint32_t lane_base = inc - sum;
        if (tid == 63) L.total = inc;
#pragma unroll
        for (int q = 0; q < MAX_BINS_PER_LANE; q++) {
            g[q] = 0;
            if (q < m && first + q < nbins) {
                const uint32_t c = L.cnt[first + q];
                L.cnt[first + q] = 0;
                L.off[first + q] += lane_base;
                if (c) g[q] = atomicAdd(&cursor[(uint64_t)(first + q) * cursor_stride], c);
            }
        }
    }
    __syncthreads();                                     // B
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++)
        if (rk[j] != NO_RANK) L.stage[L.off[bin_of(hs[j])] + rk[j]] = hs[j];
    if (w0) {
#pragma unroll
        for (int q = 0; q < MAX_BINS_PER_LANE; q++)
            if (q < m && first + q < nbins) L.delta[first + q] = g[q] - L.off[first + q];
    }
    __syncthreads();                                     // C
    const uint32_t total = L.total;
    for (uint32_t i = tid; i < total; i += THREADS) {
        const uint64_t h = L.stage[i];
        const uint32_t bin = bin_of(h);
        const uint32_t idx = i + L.delta[bin];
        if (idx < cap) out[(bin_base + bin) * cap + idx] = h;
        else n_overflow++;
    }
    if (tid == 0) n_records += total;
}

// ---- level 1: packed stream -> records partitioned by the top b1 bits of the hash ---------------
// Thread t of a tile owns the 8 consecutive positions base + 8t .. 8t+7: two bases words and two
// mask words (prefetched from HBM one tile ahead, straight to registers) cover all 8 windows,
// which are produced by shifting one 128-bit register pair; the reverse complement rolls.
template <int THREADS, int PER_THREAD, int MIN_WAVES>
__global__ void __launch_bounds__(THREADS, MIN_WAVES)
scan_part_kernel(StreamView s, int k, int canonical, uint64_t seed, int b1, uint32_t cap,
                 uint64_t *__restrict__ out, uint32_t *cursor, int cursor_stride, uint32_t n_tiles,
                 Counters *ctr)
{
    constexpr int TILE = THREADS * PER_THREAD;
    static_assert(PER_THREAD % 8 == 0 && PER_THREAD <= 16, "a thread's positions must stay inside two bases words");
    __shared__ SplitLds<THREADS, PER_THREAD> L;
    const int tid = (int)threadIdx.x;
    const int nbins = 1 << b1;
    const int shift = 64 - b1;
    auto bin_of = [=](uint64_t h) -> uint32_t { return b1 ? (uint32_t)(h >> shift) : 0u; };
    uint32_t n_records = 0, n_overflow = 0;
    for (int i = tid; i < nbins; i += THREADS) L.cnt[i] = 0;
    __syncthreads();

    const uint64_t last_b = s.n_bwords - 1, last_m = s.n_mwords - 1;
    auto load_words = [&](uint32_t tile, uint64_t &w0, uint64_t &w1, uint64_t &m0, uint64_t &m1) {
        const uint64_t p0 = (uint64_t)tile * TILE + (uint64_t)tid * PER_THREAD;
        const uint64_t bw = p0 >> 5, mw = p0 >> 6;
        w0 = s.bases[bw < last_b ? bw : last_b];
        w1 = s.bases[bw + 1 < last_b ? bw + 1 : last_b];
        m0 = s.mask[mw < last_m ? mw : last_m];
        m1 = s.mask[mw + 1 < last_m ? mw + 1 : last_m];
    };
    const int sk = 64 - 2 * k;
    const uint64_t kmask_shift = 64 - k;
    uint64_t nw0 = 0, nw1 = 0, nm0 = 0, nm1 = 0;
    if (blockIdx.x < n_tiles) load_words(blockIdx.x, nw0, nw1, nm0, nm1);

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t w0 = nw0, w1 = nw1, m0 = nm0, m1 = nm1;
        if (tile + gridDim.x < n_tiles) load_words(tile + gridDim.x, nw0, nw1, nm0, nm1);
        const uint64_t p0 = (uint64_t)tile * TILE + (uint64_t)tid * PER_THREAD;
        // left-align the stream at p0: 64 bases in (vh, vl), 64 flags in mv
        const int o = 2 * (int)(p0 & 31);                 // PER_THREAD 8: 0,16,32,48; 16: 0,32
        const uint64_t vh = o ? (w0 << o) | (w1 >> (64 - o)) : w0;
        const uint64_t vl = o ? (w1 << o) : w1;
        const int ms = (int)(p0 & 63);
        const uint64_t mv = ms ? (m0 << ms) | (m1 >> (64 - ms)) : m0;

        uint64_t hs[PER_THREAD];
        uint32_t rk[PER_THREAD];
        uint64_t rc = 0;
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            const uint64_t win = j ? (vh << (2 * j)) | (vl >> (64 - 2 * j)) : vh;
            const uint64_t fwd = win >> sk;
            if (j == 0) rc = (~rev_pairs64(fwd)) >> sk;
            else rc = (rc >> 2) | ((uint64_t)(3u - (uint32_t)(fwd & 3)) << (2 * k - 2));
            const bool bad = ((mv << j) >> kmask_shift) != 0;
            uint64_t km = fwd;
            if (canonical && rc < fwd) km = rc;
            rk[j] = NO_RANK;
            hs[j] = 0;
            if (!bad && p0 + j < s.n_bases) {
                hs[j] = fmix64(km ^ seed);
                rk[j] = atomicAdd(&L.cnt[bin_of(hs[j])], 1u);
            }
        }
        __syncthreads();                                 // A
        multisplit_flush(L, hs, rk, nbins, bin_of, cursor, cursor_stride, 0, cap, out, n_records, n_overflow);
    }
    if (tid == 0 && n_records) atomicAdd(&ctr->n_valid, (unsigned long long)n_records);
    n_overflow = (uint32_t)wave_sum(n_overflow);
    if (lane_id() == 0 && n_overflow) atomicAdd(&ctr->n_overflow, (unsigned long long)n_overflow);
}

// ---- level 2: records of coarse bin blockIdx.y -> segments by the next b2 bits --------------------
__global__ void __launch_bounds__(PART_THREADS, 8)
repart_kernel(const uint64_t *__restrict__ in, const uint32_t *__restrict__ cursor1, uint32_t cap1,
              int b1, int b2, uint32_t cap2, uint64_t *__restrict__ out, uint32_t *cursor2, Counters *ctr)
{
    __shared__ SplitLds<PART_THREADS, PART_PER_THREAD> L;
    const int tid = (int)threadIdx.x;
    const uint32_t c = blockIdx.y;
    uint32_t n = cursor1[(uint64_t)c * CURSOR_STRIDE];
    if (n > cap1) n = cap1;
    const uint32_t t0 = blockIdx.x * PART_TILE;
    if (t0 >= n) return;
    const int nbins = 1 << b2;
    const int shift = 64 - b1 - b2;
    auto bin_of = [=](uint64_t h) -> uint32_t { return (uint32_t)(h >> shift) & (uint32_t)(nbins - 1); };
    for (int i = tid; i < nbins; i += PART_THREADS) L.cnt[i] = 0;
    __syncthreads();
    const uint64_t *src = in + (uint64_t)c * cap1;
    uint64_t hs[PART_PER_THREAD];
    uint32_t rk[PART_PER_THREAD];
#pragma unroll
    for (int j = 0; j < PART_PER_THREAD; j++) {
        const uint32_t i = t0 + (uint32_t)j * PART_THREADS + tid;
        hs[j] = i < n ? src[i] : 0;
    }
#pragma unroll
    for (int j = 0; j < PART_PER_THREAD; j++) {
        const uint32_t i = t0 + (uint32_t)j * PART_THREADS + tid;
        rk[j] = NO_RANK;
        if (i < n) rk[j] = atomicAdd(&L.cnt[bin_of(hs[j])], 1u);
    }
    __syncthreads();                                     // A
    uint32_t n_records = 0, n_overflow = 0;
    multisplit_flush(L, hs, rk, nbins, bin_of, cursor2 + ((uint64_t)c << b2), 1, (uint64_t)c << b2, cap2, out,
                     n_records, n_overflow);
    n_overflow = (uint32_t)wave_sum(n_overflow);
    if (lane_id() == 0 && n_overflow) atomicAdd(&ctr->n_overflow, (unsigned long long)n_overflow);
}

// ---- per-segment kernels ----------------------------------------------------------------------------
__device__ __forceinline__ void load_segment(uint32_t *seg, const unsigned long long *filter, uint64_t seg_id)
{
    const uint4 *src = (const uint4 *)filter + seg_id * (SEG_BYTES / 16);
    uint4 *dst = (uint4 *)seg;
    for (int i = (int)threadIdx.x; i < SEG_BYTES / 16; i += (int)blockDim.x) dst[i] = src[i];
}

__global__ void __launch_bounds__(SEG_THREADS)
seg_insert_kernel(unsigned long long *filter, const uint64_t *__restrict__ recs,
                  const uint32_t *__restrict__ cursor2, uint32_t cap2, int n_hashes, int blk_shift)
{
    __shared__ __attribute__((aligned(16))) uint32_t seg[SEG_WORDS32];
    const uint64_t seg_id = blockIdx.x;
    uint32_t n = cursor2[seg_id];
    if (n == 0) return;                       // nothing to add: leave the segment untouched
    if (n > cap2) n = cap2;
    load_segment(seg, filter, seg_id);
    __syncthreads();
    const uint64_t *src = recs + seg_id * cap2;
    for (uint32_t i = threadIdx.x; i < n; i += SEG_THREADS) {
        const uint64_t h = src[i];
        const uint32_t blk = (uint32_t)(h >> blk_shift) & (SEG_BLOCKS - 1);
        const uint32_t a = (uint32_t)(h & 511), d = (uint32_t)((h >> 9) & 511) | 1u;
        for (int j = 0; j < n_hashes; j++) {
            const uint32_t bit = (a + (uint32_t)j * d) & 511;
            atomicOr(&seg[blk * 16 + (bit >> 5)], 1u << (bit & 31));
        }
    }
    __syncthreads();
    uint4 *dst = (uint4 *)filter + seg_id * (SEG_BYTES / 16);
    const uint4 *s4 = (const uint4 *)seg;
    for (int i = (int)threadIdx.x; i < SEG_BYTES / 16; i += SEG_THREADS) dst[i] = s4[i];
}

// absent records of segment s are written to miss[s * cap2 ...], their number to miss_cnt[s]
__global__ void __launch_bounds__(SEG_THREADS)
seg_probe_kernel(const unsigned long long *__restrict__ filter, const uint64_t *__restrict__ recs,
                 const uint32_t *__restrict__ cursor2, uint32_t cap2, int n_hashes, int blk_shift,
                 uint64_t *__restrict__ miss, uint32_t *__restrict__ miss_cnt, Counters *ctr)
{
    __shared__ __attribute__((aligned(16))) uint32_t seg[SEG_WORDS32];
    __shared__ uint32_t n_miss;
    const uint64_t seg_id = blockIdx.x;
    uint32_t n = cursor2[seg_id];
    if (n > cap2) n = cap2;
    if (n == 0) {
        if (threadIdx.x == 0) miss_cnt[seg_id] = 0;
        return;
    }
    if (threadIdx.x == 0) n_miss = 0;
    load_segment(seg, filter, seg_id);
    __syncthreads();
    const uint64_t *src = recs + seg_id * cap2;
    uint64_t *dst = miss + seg_id * cap2;
    const uint32_t n_round = (n + 63) & ~63u;
    for (uint32_t i = threadIdx.x; i < n_round; i += SEG_THREADS) {
        const bool have = i < n;
        const uint64_t h = have ? src[i] : 0;
        const uint32_t blk = (uint32_t)(h >> blk_shift) & (SEG_BLOCKS - 1);
        const uint32_t a = (uint32_t)(h & 511), d = (uint32_t)((h >> 9) & 511) | 1u;
        bool all = true;
        for (int j = 0; j < n_hashes; j++) {
            const uint32_t bit = (a + (uint32_t)j * d) & 511;
            all = all && ((seg[blk * 16 + (bit >> 5)] >> (bit & 31)) & 1u);
        }
        const bool absent = have && !all;
        const uint64_t b = __ballot(absent);
        if (b) {
            const int leader = __ffsll((long long)b) - 1;
            uint32_t wbase = 0;
            if (lane_id() == leader) wbase = atomicAdd(&n_miss, (uint32_t)__popcll(b));
            wbase = __shfl(wbase, leader);
            if (absent) dst[wbase + popc_below(b)] = h;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        miss_cnt[seg_id] = n_miss;
        if (n_miss) atomicAdd(&ctr->n_absent, (unsigned long long)n_miss);
    }
}

// exact counting of one segment's absent records in an LDS hash table, `rounds` sub-ranges at a time
__global__ void __launch_bounds__(CNT_THREADS)
seg_count_kernel(const uint64_t *__restrict__ list, const uint32_t *__restrict__ list_cnt, uint32_t cap2,
                 int T, uint64_t seed, uint32_t min_count, uint64_t out_cap,
                 uint64_t *__restrict__ out_kmer, uint32_t *__restrict__ out_cnt, Counters *ctr)
{
    __shared__ unsigned long long keys[CNT_SLOTS];
    __shared__ uint32_t cnts[CNT_SLOTS];
    __shared__ uint32_t wave_sums[CNT_THREADS / 64];
    __shared__ uint32_t total;
    __shared__ unsigned long long gbase;
    const uint64_t seg_id = blockIdx.x;
    uint32_t n = list_cnt[seg_id];
    if (n > cap2) n = cap2;
    if (n == 0) return;
    const uint64_t *src = list + seg_id * cap2;
    // a value no record of this segment can take: its top T bits differ from the segment id
    const unsigned long long EMPTY = (unsigned long long)(seg_id ^ 1ULL) << (64 - T);
    const uint32_t rounds = (n + CNT_SLOTS / 2 - 1) / (CNT_SLOTS / 2);
    uint32_t n_distinct = 0, n_fail = 0;
    for (uint32_t r = 0; r < rounds; r++) {
        for (int i = (int)threadIdx.x; i < CNT_SLOTS; i += CNT_THREADS) { keys[i] = EMPTY; cnts[i] = 0; }
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += CNT_THREADS) {
            const uint64_t h = src[i];
            const uint32_t rr = (uint32_t)((((h >> 33) & 0x1FFF) * rounds) >> 13);
            if (rr != r) continue;
            uint32_t slot = (uint32_t)(h >> 20) & (CNT_SLOTS - 1);
            int tries = 0;
            for (; tries < CNT_SLOTS; tries++) {
                const unsigned long long prev = atomicCAS(&keys[slot], EMPTY, (unsigned long long)h);
                if (prev == EMPTY || prev == h) { atomicAdd(&cnts[slot], 1u); break; }
                slot = (slot + 1) & (CNT_SLOTS - 1);
            }
            if (tries == CNT_SLOTS) n_fail++;       // table full: host falls back to the direct family
        }
        __syncthreads();
        // emit: count, scan, reserve, write
        uint32_t mine = 0;
        for (int s = (int)threadIdx.x; s < CNT_SLOTS; s += CNT_THREADS) {
            if (keys[s] != EMPTY) { n_distinct++; if (cnts[s] >= min_count) mine++; }
        }
        const uint32_t ex = block_excl_scan(mine, wave_sums, &total);
        if (threadIdx.x == 0) gbase = total ? atomicAdd(&ctr->n_emitted, (unsigned long long)total) : 0ULL;
        __syncthreads();
        uint64_t o = gbase + ex;
        for (int s = (int)threadIdx.x; s < CNT_SLOTS; s += CNT_THREADS) {
            if (keys[s] != EMPTY && cnts[s] >= min_count) {
                if (o < out_cap) {
                    out_kmer[o] = unfmix64(keys[s]) ^ seed;
                    out_cnt[o] = cnts[s];
                }
                o++;
            }
        }
        __syncthreads();
    }
    n_distinct = (uint32_t)wave_sum(n_distinct);
    n_fail = (uint32_t)wave_sum(n_fail);
    if (lane_id() == 0) {
        if (n_distinct) atomicAdd(&ctr->n_distinct, (unsigned long long)n_distinct);
        if (n_fail) atomicAdd(&ctr->n_overflow, (unsigned long long)n_fail);
    }
}

// ---- host side -----------------------------------------------------------------------------------------
// Records per bin are not Poisson: a k-mer seen m times (coverage, repeats) puts all m copies in
// one bin, so the variance is mean * E[m^2]/E[m].  The slack covers 8 sigma for a multiplicity
// ratio of 64 (30-60x coverage); anything heavier (poly-A style heavy hitters) overflows and is
// handled exactly by the direct family.
inline uint32_t bin_capacity(double mean)
{
    const double c = mean + 64.0 * sqrt(mean + 1.0) + 256.0;
    return (uint32_t)((uint64_t)(c + 1.0) + 1) & ~1u;
}

inline bool make_plan(const dk_engine *e, const dk_reads *r, BucketPlan *p)
{
    if (e->cfg.k > 32) return false;
    p->T = (int)e->cfg.filter_log2_bits - 9 - SEG_LOG2_BLOCKS;
    if (p->T < 1 || p->T > 2 * MAX_BIN_BITS) return false;
    if (p->T <= MAX_BIN_BITS) { p->b1 = p->T; p->b2 = 0; }
    else { p->b1 = (p->T + 1) / 2; p->b2 = p->T - p->b1; }
    p->p1 = 1u << p->b1;
    p->p2 = 1u << p->b2;
    p->n_seg = 1ULL << p->T;
    p->n_max = r->n_windows && r->n_windows < r->n_bases ? r->n_windows : r->n_bases;
    const double c1 = (double)p->n_max / p->p1, c2 = (double)p->n_max / (double)p->n_seg;
    if (c1 + 64.0 * sqrt(c1 + 1.0) + 512.0 >= 4.0e9) return false;      // u32 cursors
    p->cap1 = bin_capacity(c1);
    p->cap2 = bin_capacity(c2);
    return true;
}

// AUTO mode: bucketed when sweeping the filter once costs less than one random 64-B block per k-mer
inline bool bucketed_pays(const dk_engine *e, uint64_t n_bases)
{
    const uint64_t filter_bytes = (1ULL << e->cfg.filter_log2_bits) / 8;
    const int T = (int)e->cfg.filter_log2_bits - 9 - SEG_LOG2_BLOCKS;
    if (e->cfg.k > 32 || T < 1 || T > 2 * MAX_BIN_BITS) return false;
    return filter_bytes >= (32ULL << 20) && n_bases * 16 >= filter_bytes;
}

struct BucketBufs {
    uint64_t *a = nullptr, *b = nullptr;      // level-1 bins (later: absent lists) / segment bins
    uint32_t *cur = nullptr;                  // cursor1 [p1 * stride] | cursor2 [n_seg] | miss_cnt [n_seg]
    uint32_t *cursor1 = nullptr, *cursor2 = nullptr, *miss_cnt = nullptr;
};

inline void free_bufs(dk_engine *e, BucketBufs &B)
{
    pool_free(e, B.a);
    pool_free(e, B.b);
    pool_free(e, B.cur);
}

// scan_part (+ repart): afterwards B.b holds every record of the batch grouped by segment
inline dk_status bucketed_partition(dk_engine *e, const dk_reads *r, const BucketPlan &p, BucketBufs &B)
{
    const uint64_t seg_recs = p.n_seg * (uint64_t)p.cap2;
    const uint64_t lvl1_recs = p.b2 ? (uint64_t)p.p1 * p.cap1 : 0;
    DK_TRY(pool_alloc(e, std::max(seg_recs, lvl1_recs) * 8, (void **)&B.a));
    DK_TRY(pool_alloc(e, seg_recs * 8, (void **)&B.b));
    const uint64_t n_cur = (uint64_t)p.p1 * CURSOR_STRIDE + 2 * p.n_seg;
    DK_TRY(pool_alloc(e, n_cur * 4, (void **)&B.cur));
    B.cursor1 = B.cur;
    B.cursor2 = B.cur + (uint64_t)p.p1 * CURSOR_STRIDE;
    B.miss_cnt = B.cursor2 + p.n_seg;
    DK_HIP(e, hipMemsetAsync(B.cur, 0, n_cur * 4, e->stream));

    StreamView sv;
    sv.bases = r->d_bases;
    sv.mask = r->d_mask;
    sv.n_bases = r->n_bases;
    sv.n_bwords = (r->n_bases + 31) / 32;
    sv.n_mwords = (r->n_bases + 63) / 64;
    // scan_part variants (threads x positions per thread, min waves/SIMD); DK_SCAN_VARIANT picks one
    static const int variant = [] { const char *v = getenv("DK_SCAN_VARIANT"); return v ? atoi(v) : 2; }();
    const int tile = variant == 2 ? 512 * 16 : variant == 3 ? 512 * 8 : 1024 * 8;
    const int blocks_per_cu = variant == 1 ? 1 : variant == 3 ? 4 : 2;
    const uint64_t n_tiles = (r->n_bases + tile - 1) / tile;
    if (n_tiles > 0xFFFFFFFFULL) return fail(e, DK_ERR_UNSUPPORTED, "batch too large for one bucketed pass");
    const int grid = (int)std::min<uint64_t>(n_tiles, (uint64_t)e->n_cu * blocks_per_cu);
    const bool two = p.b2 != 0;
    uint64_t *dst = two ? B.a : B.b;
    uint32_t *cur = two ? B.cursor1 : B.cursor2;
    const int stride = two ? CURSOR_STRIDE : 1;
    const uint32_t cap = two ? p.cap1 : p.cap2;
#define DK_SCAN_LAUNCH(T, P, W)                                                                              \
    scan_part_kernel<T, P, W><<<grid, T, 0, e->stream>>>(sv, (int)e->cfg.k, (int)e->cfg.canonical, e->cfg.seed, \
                                                         p.b1, cap, dst, cur, stride, (uint32_t)n_tiles, e->d_ctr)
    switch (variant) {
    case 1: DK_SCAN_LAUNCH(1024, 8, 4); break;
    case 2: DK_SCAN_LAUNCH(512, 16, 4); break;
    case 3: DK_SCAN_LAUNCH(512, 8, 8); break;
    default: DK_SCAN_LAUNCH(1024, 8, 8); break;
    }
#undef DK_SCAN_LAUNCH
    DK_HIP(e, hipGetLastError());
    stage_mark(e, "scan_part");
    if (two) {
        const dim3 g2((p.cap1 + PART_TILE - 1) / PART_TILE, p.p1);
        repart_kernel<<<g2, PART_THREADS, 0, e->stream>>>(B.a, B.cursor1, p.cap1, p.b1, p.b2, p.cap2, B.b, B.cursor2, e->d_ctr);
        DK_HIP(e, hipGetLastError());
        stage_mark(e, "repart");
    }
    return DK_OK;
}

// Returns DK_ERR_OVERFLOW (without touching e->err semantics beyond the message) when a bin
// overflowed: the caller then runs the direct family on the whole batch, which is exact (OR is
// idempotent, so records already inserted do no harm).
inline dk_status bucketed_insert(dk_engine *e, dk_set *s, const dk_reads *r)
{
    BucketPlan p;
    if (!make_plan(e, r, &p)) return fail(e, DK_ERR_UNSUPPORTED, "no bucketed plan for this geometry");
    BucketBufs B;
    dk_status st = bucketed_partition(e, r, p, B);
    if (st == DK_OK) {
        seg_insert_kernel<<<(unsigned)p.n_seg, SEG_THREADS, 0, e->stream>>>(
            s->d_words, B.b, B.cursor2, p.cap2, (int)e->cfg.n_hashes, 64 - p.T - SEG_LOG2_BLOCKS);
        hipError_t h = hipGetLastError();
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_insert launch failed: %s", hipGetErrorString(h));
        else stage_mark(e, "seg_insert");
    }
    if (st == DK_OK) {
        hipError_t h = hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "bucketed insert failed: %s", hipGetErrorString(h));
        else if (e->h_ctr->n_overflow) st = fail(e, DK_ERR_OVERFLOW, "bucket overflow (%llu records)", (unsigned long long)e->h_ctr->n_overflow);
    }
    free_bufs(e, B);
    return st;
}

inline dk_status bucketed_probe(dk_engine *e, dk_set *s, const dk_reads *r, dk_result *res)
{
    BucketPlan p;
    if (!make_plan(e, r, &p)) return fail(e, DK_ERR_UNSUPPORTED, "no bucketed plan for this geometry");
    BucketBufs B;
    dk_status st = bucketed_partition(e, r, p, B);
    auto sync_counters = [&]() -> dk_status {
        hipError_t h = hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) return fail(e, DK_ERR_HIP, "bucketed probe failed: %s", hipGetErrorString(h));
        if (e->h_ctr->n_overflow) return fail(e, DK_ERR_OVERFLOW, "bucket overflow (%llu records)", (unsigned long long)e->h_ctr->n_overflow);
        return DK_OK;
    };
    const uint64_t *list = B.b;
    const uint32_t *list_cnt = B.cursor2;
    if (st == DK_OK && s) {
        seg_probe_kernel<<<(unsigned)p.n_seg, SEG_THREADS, 0, e->stream>>>(
            s->d_words, B.b, B.cursor2, p.cap2, (int)e->cfg.n_hashes, 64 - p.T - SEG_LOG2_BLOCKS, B.a, B.miss_cnt, e->d_ctr);
        hipError_t h = hipGetLastError();
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_probe launch failed: %s", hipGetErrorString(h));
        else stage_mark(e, "seg_probe");
        list = B.a;
        list_cnt = B.miss_cnt;
    }
    if (st == DK_OK) st = sync_counters();
    uint64_t n_absent = 0;
    if (st == DK_OK) {
        if (!s) {            // KmerCounter semantics: every valid k-mer is counted
            e->h_ctr->n_absent = e->h_ctr->n_valid;
            hipError_t h = hipMemcpyAsync(&e->d_ctr->n_absent, &e->h_ctr->n_valid, 8, hipMemcpyHostToDevice, e->stream);
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "counter update failed: %s", hipGetErrorString(h));
        }
        n_absent = e->h_ctr->n_absent;
    }
    if (st == DK_OK && n_absent) {
        st = pool_alloc(e, n_absent * 8, (void **)&res->d_lo);
        if (st == DK_OK) st = pool_alloc(e, n_absent * 4, (void **)&res->d_cnt);
        if (st == DK_OK) {
            seg_count_kernel<<<(unsigned)p.n_seg, CNT_THREADS, 0, e->stream>>>(
                list, list_cnt, p.cap2, p.T, e->cfg.seed, e->cfg.min_count, n_absent, res->d_lo, res->d_cnt, e->d_ctr);
            hipError_t h = hipGetLastError();
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_count launch failed: %s", hipGetErrorString(h));
            else stage_mark(e, "seg_count");
        }
        if (st == DK_OK) st = sync_counters();
        if (st == DK_OK) res->n = e->h_ctr->n_emitted;
    }
    free_bufs(e, B);
    return st;
}

}  // namespace dk
