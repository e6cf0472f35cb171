// dk_bucket_common.h -- records, overflow list, block scans and the LDS multisplit shared by the bucketed kernels
// (part of the bucketed kernel family: dk_kernels_bucket.h has the overview and includes the parts in order)
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
constexpr int MAX_BIN_BITS = 9;                        // 512-thread multisplit kernels (the bin scan is one thread per bin): k > 32, small geometries
constexpr int MAX_BINS = 1 << MAX_BIN_BITS;
constexpr int MAX_BIN_BITS1 = 10;                      // level 1 of the 1024-thread scan_part (16 K-record tiles: runs of 16 records)
constexpr int MAX_BIN_BITS2 = 10;                      // later levels (repart: one bin per XCD, its L2 assembles the lines of short runs)
constexpr int MAX_BINS2 = 1 << MAX_BIN_BITS2;
constexpr int MAX_SEG_BITS = 23;                       // two levels up to 18 bits, three levels beyond (coarse regions <= 2^15: grid y)
constexpr int CURSOR_STRIDE = 32;                      // level-1 cursors on separate 128-B lines

constexpr int SEG_THREADS = 1024;
// The set kernels keep one 64-KiB segment in LDS and run two workgroups of 1024 threads per CU, i.e. 8 waves per SIMD:
// that needs at most 64 VGPRs AND at most 80 SGPRs per wave -- the CU admits floor(800 / (ceil(sgpr / 16) * 16 + 16))
// waves per SIMD (MI355X_MICROARCH.md, Residency), and a kernel at 87 SGPRs silently ran one workgroup per CU
// (seg_probe 9.6 -> 13.2 ms at 2^39 bits).  `make resources` prints what the compiler settled on.
#define DK_SEG_KERNEL __global__ void __launch_bounds__(SEG_THREADS, 8) __attribute__((amdgpu_num_sgpr(72)))
constexpr uint32_t NO_RANK = 0xFFFFFFFFu;

constexpr int MAX_R = 8;                               // pieces per counting unit: adjacent segments counted together, or the ranks of a multi-GPU run

// Level-1 buckets are built from PRIVATE pieces, so the scan needs no global atomics at all:
//   level 1: workgroup w of scan_part owns piece (bin b, w) = a[(b*G + w) * capw ...]
//   level 2: one repart workgroup per tile of a level-1 piece appends to the per-segment regions
//            b[s * cap2 ...] through global cursors (shared, hot write frontiers; see repart_kernel)
// Level-1 cursors live in LDS for the life of the workgroup; piece sizes are stored once at the end.
struct BucketPlan {
    int T;                 // log2(number of segments)
    int b1, b2;            // hash bits consumed at level 1 / level 2 (b1 + b2 [+ b3] = T, b2 >= 1)
    int b3;                // > 0: a third level (2^19 segments and more): level 2 fills 2^(b1+b2) coarse regions of
                           // capA records, a second repart pass splits each by b3 more bits into the segments
    uint32_t capA;
    uint32_t p1, p2;       // bins at each level
    uint64_t n_seg;
    uint32_t G;            // scan_part workgroups = level-1 pieces per bin
    uint32_t capw, cap2;   // records per level-1 piece / per segment
    uint64_t n_max;        // upper bound on records of the batch
    int tile;              // positions per scan_part tile
    int variant;           // scan_part geometry (see make_plan)
    int sbits;             // sub-segment split: every partition region covers 2^sbits 64-KiB segments (PieceList::sbits)
    // Slab-wise level 2 (insert / accumulate, two levels): the scan writes all level-1 bins once; the bins are then taken
    // `slab_bins` at a time -- repart of the slab's bins into ONE slab-sized set of regions, the set kernel over the slab's
    // segments, next slab -- so the regions need 1 / slabs of the room (2^39 bits, 32 M reads: 40 GB -> 0.6-5 GB), which is
    // what lets a whole-genome child pass run in ONE hash window beside the filter and a full-size accumulator.
    uint32_t slabs, slab_bins;
    bool packed2;          // level-2 regions hold packed 6-byte records (k <= 32, two levels, >= 16 prefix bits per region)
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global load and store of the wave (s_waitcnt vmcnt(0)), which would serialise the prefetch of the
// next tile and the copy-out stores of the previous one behind each barrier; the partition kernels
// exchange data through LDS only, so lgkmcnt(0) + s_barrier is the required ordering.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// DK_STAMPS diagnostic build: thread 0 of each workgroup sums the cycles between phase marks and
// adds them to Counters::dbg at the end (never enabled in the shipped library)
struct Stamps {
#ifdef DK_STAMPS
    unsigned long long prev, acc[4];
    __device__ __forceinline__ Stamps() : prev(clock64()), acc{0, 0, 0, 0} {}
    __device__ __forceinline__ void mark(int i)
    {
        if (threadIdx.x == 0) { const unsigned long long t = clock64(); acc[i] += t - prev; prev = t; }
    }
    __device__ __forceinline__ void flush(Counters *ctr, int base)
    {
        if (threadIdx.x == 0) for (int i = 0; i < 4; i++) atomicAdd(&ctr->dbg[base + i], acc[i]);
    }
#else
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(Counters *, int) {}
#endif
};

// exclusive prefix sum over the block; every thread calls it; *total gets the block sum.
// LDS_ONLY: the barriers order LDS traffic only (lds_barrier), for kernels with global stores in flight
template <bool LDS_ONLY = false>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t *wave_sums, uint32_t *total)
{
    const int lane = lane_id();
    const int wave = (int)(threadIdx.x >> 6);
    const int n_waves = (int)(blockDim.x >> 6);
    const uint32_t inc = wave_incl_scan(v);
    if (lane == 63) wave_sums[wave] = inc;
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    if (wave == 0) {
        const uint32_t w = lane < n_waves ? wave_sums[lane] : 0;
        const uint32_t wi = wave_incl_scan(w);
        if (lane < n_waves) wave_sums[lane] = wi - w;       // exclusive wave offsets
        if (lane == n_waves - 1) *total = wi;
    }
    if (LDS_ONLY) lds_barrier(); else __syncthreads();
    return wave_sums[wave] + inc - v;
}

// Overflow list: records that do not fit their piece / segment region (heavy-hitter k-mers such as
// poly-A, or skew beyond the capacity slack) are appended here instead of being dropped, and are
// handled exactly afterwards: OR-ed into the filter one by one (insert), or probed one by one and
// handed to seg_count as an extra per-segment list (probe).  Only if this list overflows too is the
// batch redone by the direct family.
// Bucket records.  k <= 32: the hash alone (a bijection of the k-mer).  33 <= k <= 64: the hash of
// the low word tweaked by the high word, plus the high word: (h, hi) -> lo = unfmix64(h) ^ tweak(hi).
struct Rec1 {
    uint64_t h;
};
struct alignas(16) Rec2 {
    uint64_t h, hi;
};
template <bool WIDE> struct RecOf { using type = Rec1; };
template <> struct RecOf<true> { using type = Rec2; };

__device__ __forceinline__ uint64_t rec_hi(const Rec1 &) { return 0; }
__device__ __forceinline__ uint64_t rec_hi(const Rec2 &r) { return r.hi; }
__device__ __forceinline__ uint64_t rec_lo(const Rec1 &r, uint64_t seed) { return unfmix64(r.h) ^ seed; }
__device__ __forceinline__ uint64_t rec_lo(const Rec2 &r, uint64_t seed) { return unfmix64(r.h) ^ hash_tweak<true>(r.hi, seed); }
__device__ __forceinline__ bool rec_eq(const Rec1 &a, const Rec1 &b) { return a.h == b.h; }
__device__ __forceinline__ bool rec_eq(const Rec2 &a, const Rec2 &b) { return a.h == b.h && a.hi == b.hi; }

// a store through a pointer that was kept as an integer (LDS-resident addresses): tell the compiler it is global memory,
// or it emits a flat store, which also occupies the LDS counter the kernel's ds_* waits look at
typedef unsigned long long dk_ull2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void store_global(Rec1 *dst, const Rec1 &r)
{
    *(__attribute__((address_space(1))) unsigned long long *)dst = r.h;
}
__device__ __forceinline__ void store_global(Rec2 *dst, const Rec2 &r)
{
    dk_ull2 v;
    v.x = r.h;
    v.y = r.hi;
    *(__attribute__((address_space(1))) dk_ull2 *)dst = v;
}

// Packed records (k <= 32): where all records of a group share at least 16 leading hash bits -- the counting units of an
// accumulator (T + u bits: every accumulator of a set of 2^28 bits or more) and the level-2 regions of a partition
// (window + b1 + b2 bits: sets of 2^35 bits or more) -- only the low 48 bits are kept: 6 bytes per record instead of 8.
// A group of `cap` records (a multiple of 64) is a row of 384-byte blocks, each holding 64 records as 64 x u32 (hash bits
// 0..31) followed by 64 x u16 (bits 32..47): everything stays naturally aligned, a wave's 64 consecutive records are one
// block (two fully coalesced accesses), and -- what decides the speed of the kernels that APPEND to many groups at once --
// a group has ONE write frontier of three cache lines.  (Two separate arrays per group, cap x u32 then cap x u16, gave
// every group two frontiers tens of KB apart: repart 21.5 -> 26.3 ms on the whole-genome child step.)
constexpr int PACKED_REC_BYTES = 6;
constexpr int PACKED_MIN_PREFIX_BITS = 16;
constexpr int PACKED_BLOCK_RECS = 64, PACKED_BLOCK_BYTES = PACKED_BLOCK_RECS * PACKED_REC_BYTES;
__device__ __forceinline__ void packed_store(void *store, uint64_t unit, uint32_t cap, uint32_t pos, uint64_t h)
{
    char *blk = (char *)store + unit * (uint64_t)cap * PACKED_REC_BYTES + (uint64_t)(pos >> 6) * PACKED_BLOCK_BYTES;
    ((uint32_t *)blk)[pos & 63] = (uint32_t)h;
    ((uint16_t *)(blk + PACKED_BLOCK_RECS * 4))[pos & 63] = (uint16_t)(h >> 32);
}
__device__ __forceinline__ uint64_t packed_load(const void *store, uint64_t unit, uint32_t cap, uint32_t pos, uint64_t prefix)
{
    const char *blk = (const char *)store + unit * (uint64_t)cap * PACKED_REC_BYTES + (uint64_t)(pos >> 6) * PACKED_BLOCK_BYTES;
    return prefix | ((uint64_t)((const uint16_t *)(blk + PACKED_BLOCK_RECS * 4))[pos & 63] << 32) | ((const uint32_t *)blk)[pos & 63];
}

template <class R>
struct OvfList {
    R *recs;
    unsigned long long *count;     // &Counters::n_ovf
    uint64_t cap;
};

template <class R>
__device__ __forceinline__ void ovf_append(const OvfList<R> &ovf, bool pred, const R &rec, uint32_t &n_dropped)
{
    const uint64_t slot = wave_append(pred, ovf.count);
    if (pred) {
        if (slot < ovf.cap) ovf.recs[slot] = rec;
        else n_dropped++;
    }
}

// ---- shared multisplit tail ------------------------------------------------------------------------
// A tile's records sit in registers (hs = hash, rk = rank inside its bin from the LDS count).
// Wave 0 turns the per-bin counts into tile offsets and advances the workgroup's running cursors
// (all in LDS); every wave then scatters its records into the LDS stage and the stage is copied
// out as per-bin runs.  Three barriers per tile (A: counts done - by the caller, B: offsets ready,
// C: stage ready); the next tile's count phase needs no barrier because it touches only cnt[],
// which wave 0 re-zeroes before B.
template <int THREADS, int PER_THREAD, class R, int NB = (THREADS >= 1024 ? 1024 : MAX_BINS), bool PRIVATE = true>
struct SplitLds {
    static constexpr int N_BINS = NB;
    R stage[THREADS * PER_THREAD];
    uint32_t cnt[NB];           // per-tile counts; zero on entry to every count phase
    uint32_t off[NB];           // tile offset of each bin in stage[]
    uint32_t delta[NB];         // index in the piece = stage index + delta[bin]  (mod 2^32)
    uint32_t cur[PRIVATE ? NB : 1];              // scan_part: running fill of this workgroup's piece of each bin
    unsigned long long gptr[PRIVATE ? NB : 1];   // scan_part: byte address of (piece slot of stage index 0) per bin
    uint32_t total;
    uint32_t ovf_seen;          // some bin of this workgroup has run past its capacity (never cleared)
    uint32_t ovf_lost;          // scan_part: records that found the overflow list full (kept here, not in a register of the hot loop)
};


template <int THREADS, int PER_THREAD, class R>
__device__ __forceinline__ void multisplit_init(SplitLds<THREADS, PER_THREAD, R> &L, int nbins)
{
    for (int i = (int)threadIdx.x; i < L.N_BINS; i += THREADS) { L.cnt[i] = 0; L.cur[i] = 0; }
    if (threadIdx.x == 0) { L.ovf_seen = 0; L.ovf_lost = 0; }
    __syncthreads();
}

// piece sizes, once per workgroup: cnt_out[(bin_base + bin) * n_pieces + piece] = min(fill, cap)
template <int THREADS, int PER_THREAD, class R>
__device__ __forceinline__ void multisplit_finish(SplitLds<THREADS, PER_THREAD, R> &L, int nbins, uint64_t bin_base,
                                                  uint32_t n_pieces, uint32_t piece, uint32_t cap,
                                                  uint32_t *__restrict__ cnt_out)
{
    __syncthreads();
    for (int i = (int)threadIdx.x; i < nbins; i += THREADS) {
        const uint32_t c = L.cur[i];
        cnt_out[(bin_base + i) * n_pieces + piece] = c < cap ? c : cap;
    }
}

}  // namespace dk
