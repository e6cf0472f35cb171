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
constexpr int MAX_BIN_BITS = 9;                        // level 1 (scan_part: private pieces, the runs must stay long)
constexpr int MAX_BINS = 1 << MAX_BIN_BITS;
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
template <int THREADS, int PER_THREAD, class R, int NB = MAX_BINS, bool PRIVATE = true>
struct SplitLds {
    R stage[THREADS * PER_THREAD];
    uint32_t cnt[NB];           // per-tile counts; zero on entry to every count phase
    uint32_t off[NB];           // tile offset of each bin in stage[]
    uint32_t delta[NB];         // index in the piece = stage index + delta[bin]  (mod 2^32)
    uint32_t cur[PRIVATE ? NB : 1];              // scan_part: running fill of this workgroup's piece of each bin
    unsigned long long gptr[PRIVATE ? NB : 1];   // scan_part: byte address of (piece slot of stage index 0) per bin
    uint32_t total;
    uint32_t ovf_seen;          // some bin of this workgroup has run past its capacity (never cleared)
};


template <int THREADS, int PER_THREAD, class R>
__device__ __forceinline__ void multisplit_init(SplitLds<THREADS, PER_THREAD, R> &L, int nbins)
{
    for (int i = (int)threadIdx.x; i < MAX_BINS; i += THREADS) { L.cnt[i] = 0; L.cur[i] = 0; }
    if (threadIdx.x == 0) L.ovf_seen = 0;
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

// ---- level 1: packed stream -> records partitioned by the top b1 bits of the hash ---------------
// Thread t of a tile owns PER_THREAD consecutive positions: two (k > 32: three) bases words and two
// mask words, prefetched from HBM one tile ahead straight to registers, cover all its windows, which
// are produced by shifting one register group; the reverse complement rolls.
//
// Per tile: count (hash every window, LDS atomic gives its rank in its bin) | A | scan (lane = bin,
// DPP; advances the workgroup's private cursors) | B | scatter into the LDS stage | C | copy-out of
// the stage as per-bin runs.  The copy-out of tile i (LDS reads + global stores) is interleaved,
// record by record, with the count phase of tile i+1 (pure VALU + one LDS atomic), so the LDS and
// store latency of one hides behind the hashing of the other inside every wave.
// WINDOWED: only the k-mers whose hash starts with the wbits (>= 1) bits of widx become records (a hash-range pass of
// dk_accum_add); the bins are then taken from the b1 bits after the window's.
template <int THREADS, int PER_THREAD, int MIN_WAVES, bool WIDE, bool WINDOWED>
__global__ void __launch_bounds__(THREADS, MIN_WAVES)
scan_part_kernel(StreamView s, int k, int canonical, uint64_t seed, int b1, uint32_t capw,
                 typename RecOf<WIDE>::type *__restrict__ out, uint32_t *__restrict__ cnt1, uint32_t n_tiles,
                 OvfList<typename RecOf<WIDE>::type> ovf, Counters *ctr, int wbits, uint32_t widx, uint32_t bin_skew)
{
    using R = typename RecOf<WIDE>::type;
    constexpr int TILE = THREADS * PER_THREAD;
    static_assert(PER_THREAD % 8 == 0 && PER_THREAD <= (WIDE ? 8 : 16), "a thread's positions must stay inside its bases words");
    static_assert(THREADS * PER_THREAD <= 65536, "ranks are kept in 16 bits");
    __shared__ SplitLds<THREADS, PER_THREAD, R> L;
    const int tid = (int)threadIdx.x;
    const int nbins = 1 << b1;
    // the bin and the window come from the top 32 bits of the hash (b1 + wbits <= 32): one v_bfe_u32 / one 32-bit shift
    // instead of a 64-bit shift and a mask
    const uint32_t bshift = (uint32_t)(32 - b1 - (WINDOWED ? wbits : 0));
    const uint32_t wshift = (uint32_t)(32 - wbits);      // WINDOWED only (wbits >= 1)
    uint32_t b1_v;                                       // the field width, kept in a vector register (one scalar operand per instruction)
    asm volatile("v_mov_b32 %0, %1" : "=v"(b1_v) : "s"((uint32_t)b1));
    auto bin_of = [=](uint64_t h) -> uint32_t { return __builtin_amdgcn_ubfe((uint32_t)(h >> 32), bshift, b1_v); };
    const uint64_t canon_mask = canonical ? ~0ULL : 0ULL;
    uint32_t n_records = 0, n_overflow = 0;
    uint32_t n_all = 0;                                   // WINDOWED: valid windows inside or outside the window
    multisplit_init(L, nbins);
    Stamps st;

    const uint64_t last_b = s.n_bwords - 1, last_m = s.n_mwords - 1;
    auto load_words = [&](uint32_t tile, uint64_t &w0, uint64_t &w1, uint64_t &w2, uint64_t &m0, uint64_t &m1) {
        const uint64_t p0 = (uint64_t)tile * TILE + (uint64_t)tid * PER_THREAD;
        const uint64_t bw = p0 >> 5, mw = p0 >> 6;
        w0 = s.bases[bw < last_b ? bw : last_b];
        w1 = s.bases[bw + 1 < last_b ? bw + 1 : last_b];
        w2 = WIDE ? s.bases[bw + 2 < last_b ? bw + 2 : last_b] : 0;
        m0 = s.mask[mw < last_m ? mw : last_m];
        m1 = s.mask[mw + 1 < last_m ? mw + 1 : last_m];
    };
    const int sk = (WIDE ? 128 : 64) - 2 * k;            // right-alignment shift of a window
    const uint64_t kmask_shift = 64 - k;
    const uint64_t G = gridDim.x, w = blockIdx.x;
    // bin_skew: records between the end of one bin's pieces and the start of the next bin's (keeps the 2^b1 write frontiers
    // of a workgroup, G * capw records apart, off a common multiple of 4 KiB)
    const uint64_t piece_base = w * capw, bin_stride = G * capw + bin_skew;

    // the tile being hashed: stream left-aligned at p0 -- bases in (v0, v1[, v2]), flags in (mh, ml)
    uint64_t p0 = 0, v0 = 0, v1 = 0, v2 = 0, mh = 0, ml = 0, rch = 0, rcl = 0;
    uint32_t okbits = 0;                                   // k <= 32: bit (PER_THREAD - 1 - j) = window j is a k-mer
    auto prep = [&](uint32_t tile, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t m0, uint64_t m1) {
        p0 = (uint64_t)tile * TILE + (uint64_t)tid * PER_THREAD;
        const int o = 2 * (int)(p0 & 31);                 // PER_THREAD 8: 0,16,32,48; 16: 0,32
        v0 = o ? (w0 << o) | (w1 >> (64 - o)) : w0;
        v1 = o ? (w1 << o) | (WIDE ? w2 >> (64 - o) : 0) : w1;
        v2 = WIDE ? (o ? w2 << o : w2) : 0;
        const int ms = (int)(p0 & 63);
        mh = ms ? (m0 << ms) | (m1 >> (64 - ms)) : m0;
        ml = WIDE ? (ms ? m1 << ms : m1) : 0;
        if constexpr (!WIDE) {
            // all PER_THREAD validity flags at once: smear every mask flag over the k - 1 positions before it
            // (bit 63 - j of y = any flag in [j, j + k)), then cut at the end of the stream
            uint64_t y = mh;
            int cov = 1;
            while (cov * 2 <= k) { y |= y << cov; cov *= 2; }
            if (cov < k) y |= y << (k - cov);
            const uint32_t bad = (uint32_t)(y >> (64 - PER_THREAD));
            const uint64_t left = p0 < s.n_bases ? s.n_bases - p0 : 0;      // positions of this thread inside the stream
            const uint32_t inside = left >= (uint64_t)PER_THREAD ? (1u << PER_THREAD) - 1u
                                                                 : ~((1u << (PER_THREAD - (uint32_t)left)) - 1u) & ((1u << PER_THREAD) - 1u);
            okbits = ~bad & inside;
            if constexpr (WINDOWED) n_all += (uint32_t)__popc(okbits);
        }
    };
    // window j of the tile being hashed -> record; returns true when the window is a k-mer
    auto window = [&](int j, R &rec) -> bool {
        uint64_t kh = 0, kl;
        bool bad;
        if (!WIDE) {
            // 64 bits of the stream from base j on: two funnel shifts over (v0, top word of v1); j < 16
            const uint32_t a2 = (uint32_t)(v0 >> 32), a1 = (uint32_t)v0, a0 = (uint32_t)(v1 >> 32);
            const uint64_t win = j ? ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, 32 - 2 * j) << 32) | __builtin_amdgcn_alignbit(a1, a0, 32 - 2 * j) : v0;
            const uint64_t fwd = win >> sk;
            if (j == 0) rcl = (~rev_pairs64(fwd)) >> sk;
            else rcl = (rcl >> 2) | ((uint64_t)(3u - (uint32_t)(fwd & 3)) << (2 * k - 2));
            bad = false;                                   // decided for all windows at once in prep()
            // (the strand choice as a lane mask ANDed with the option on the scalar unit: one compare and one pair of selects)
            kl = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64(rcl < fwd) & canon_mask) ? rcl : fwd;
        } else {
            const uint64_t A = j ? (v0 << (2 * j)) | (v1 >> (64 - 2 * j)) : v0;
            const uint64_t B = j ? (v1 << (2 * j)) | (v2 >> (64 - 2 * j)) : v1;
            const uint64_t fh = sk ? A >> sk : A;
            const uint64_t fl = sk ? (B >> sk) | (A << (64 - sk)) : B;
            if (j == 0) {
                const uint64_t th = ~rev_pairs64(fl), tl = ~rev_pairs64(fh);
                rch = sk ? th >> sk : th;
                rcl = sk ? (tl >> sk) | (th << (64 - sk)) : tl;
            } else {
                rcl = (rcl >> 2) | (rch << 62);
                rch = (rch >> 2) | ((uint64_t)(3u - (uint32_t)(fl & 3)) << (2 * k - 2 - 64));
            }
            const uint64_t mx = j ? (mh << j) | (ml >> (64 - j)) : mh;
            bad = (mx >> kmask_shift) != 0;
            const bool use_rc = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64(rch < fh || (rch == fh && rcl < fl)) & canon_mask);
            kh = use_rc ? rch : fh;
            kl = use_rc ? rcl : fl;
        }
        rec.h = fmix64(kl ^ hash_tweak<WIDE>(kh, seed));
        if constexpr (WIDE) rec.hi = kh;
        bool ok;
        if constexpr (!WIDE) ok = (okbits >> (PER_THREAD - 1 - j)) & 1u;
        else ok = !bad && p0 + j < s.n_bases;
        if constexpr (WINDOWED) {
            if constexpr (WIDE) n_all += ok;                // (k <= 32: counted per tile from okbits, in prep)
            ok = ok && ((uint32_t)(rec.h >> 32) >> wshift) == widx;
        }
        return ok;
    };

    uint32_t tile = blockIdx.x;
    if (tile < n_tiles) {
        uint64_t nw0 = 0, nw1 = 0, nw2 = 0, nm0 = 0, nm1 = 0;
        {
            uint64_t w0, w1, w2, m0, m1;
            load_words(tile, w0, w1, w2, m0, m1);
            if (tile + gridDim.x < n_tiles) load_words(tile + gridDim.x, nw0, nw1, nw2, nm0, nm1);
            prep(tile, w0, w1, w2, m0, m1);
        }
        R hs[PER_THREAD];
        uint32_t rk[PER_THREAD / 2];                     // ranks are < TILE <= 2^16: two per register
        uint32_t valid = 0;
        // count phase of the first tile
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            uint32_t r = 0;
            if (window(j, hs[j])) {
                valid |= 1u << j;
                r = atomicAdd(&L.cnt[bin_of(hs[j].h)], 1u);
            }
            rk[j / 2] = (j & 1) ? rk[j / 2] | (r << 16) : r;
            // keep the windows sequential: interleaving the hash chains costs ~40 VGPRs and with
            // them half the resident waves, which hide latency better than in-wave ILP does
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll 1
        for (;;) {
            lds_barrier();                               // A: the tile's counts are complete
            st.mark(0);
            // scan of the bin counts, lane = bin; each scanning wave sums the waves below it itself
            // (independent LDS reads, one DPP reduction), so no wave waits for another
            const int wv = tid >> 6, lane = tid & 63;
            if (wv * 64 < nbins) {
                const uint32_t c = tid < nbins ? L.cnt[tid] : 0;
                const uint32_t cu = tid < nbins ? L.cur[tid] : 0;
                uint32_t below = 0;
#pragma unroll
                for (int v = 0; v < MAX_BINS / 64 - 1; v++) {
                    const uint32_t x = L.cnt[64 * v + lane];    // unconditional: the reads pipeline (cnt[] is zero beyond nbins)
                    below += v < wv ? x : 0u;
                }
                below = wave_total(below);
                const uint32_t ex = below + wave_incl_scan(c) - c;
                if (tid < nbins) {
                    L.off[tid] = ex;
                    L.delta[tid] = cu - ex;
                    // where stage slot 0 would land in this bin's piece: the copy-out adds 8 * slot
                    L.gptr[tid] = (unsigned long long)(uintptr_t)(out + ((uint64_t)tid * bin_stride + piece_base + cu - ex));
                    L.cur[tid] = cu + c;
                    if (cu + c > capw) L.ovf_seen = 1;
                    if (tid == nbins - 1) L.total = ex + c;
                }
            }
            lds_barrier();                               // B: offsets ready
            st.mark(1);
            if (tid < nbins) L.cnt[tid] = 0;             // every scanning wave has read it
#pragma unroll
            for (int j = 0; j < PER_THREAD; j++)
                if ((valid >> j) & 1u) L.stage[L.off[bin_of(hs[j].h)] + ((rk[j / 2] >> (16 * (j & 1))) & 0xffffu)] = hs[j];
            lds_barrier();                               // C: stage ready, cnt[] zero
            st.mark(2);
            const uint32_t total = L.total;
            const bool checked = L.ovf_seen != 0;        // some piece may be full: bounds check + overflow list
            const bool has_next = tile + gridDim.x < n_tiles;
            if (has_next) {
                const uint64_t w0 = nw0, w1 = nw1, w2 = nw2, m0 = nm0, m1 = nm1;
                if (tile + 2 * gridDim.x < n_tiles) load_words(tile + 2 * gridDim.x, nw0, nw1, nw2, nm0, nm1);
                prep(tile + gridDim.x, w0, w1, w2, m0, m1);
            }
            valid = 0;
            // copy-out of this tile, interleaved with the count phase of the next one
#pragma unroll
            for (int j = 0; j < PER_THREAD; j++) {
                const uint32_t i = (uint32_t)j * THREADS + tid;
                const bool mine = i < total;
                R rec;
                uint32_t bin = 0, idx = 0;
                R *dst = nullptr;
                if (mine) {
                    rec = L.stage[i];
                    bin = bin_of(rec.h);
                    if (!checked) dst = (R *)(uintptr_t)L.gptr[bin] + i;
                    else idx = i + L.delta[bin];         // 32-bit on purpose: delta is a wrapped difference
                }
                if (has_next) {
                    uint32_t r = 0;
                    if (window(j, hs[j])) {
                        valid |= 1u << j;
                        r = atomicAdd(&L.cnt[bin_of(hs[j].h)], 1u);
                    }
                    rk[j / 2] = (j & 1) ? rk[j / 2] | (r << 16) : r;
                }
                if (!checked) {
                    if (mine) store_global(dst, rec);
                } else {
                    if (mine && idx < capw) out[(uint64_t)bin * bin_stride + piece_base + idx] = rec;
                    ovf_append(ovf, mine && idx >= capw, rec, n_overflow);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (tid == 0) n_records += total;
            st.mark(3);
            if (!has_next) break;
            tile += gridDim.x;
        }
    }
    st.flush(ctr, 0);
    multisplit_finish(L, nbins, 0, (uint32_t)G, (uint32_t)w, capw, cnt1);
    if constexpr (WINDOWED) {
        if (tid == 0 && n_records) atomicAdd(&ctr->n_in_window, (unsigned long long)n_records);
        n_all = wave_total(n_all);
        if (lane_id() == 0 && n_all) atomicAdd(&ctr->n_valid, (unsigned long long)n_all);
    } else {
        if (tid == 0 && n_records) atomicAdd(&ctr->n_valid, (unsigned long long)n_records);
    }
    n_overflow = (uint32_t)wave_sum(n_overflow);
    if (lane_id() == 0 && n_overflow) atomicAdd(&ctr->n_overflow, (unsigned long long)n_overflow);
}

// ---- kmer.rs stand-in at streaming speed: canonical k-mer / hash / not-a-k-mer bit per position ------------
// Same window machinery as scan_part (a thread owns 16 consecutive positions, two or three register
// words cover all its windows, the reverse complement rolls), but nothing is partitioned: every wave
// transposes its 1024 results through its own 8.5 KiB of LDS so that each store instruction writes 64
// consecutive positions.  No workgroup barrier anywhere.
template <int THREADS, bool WIDE>
__global__ void __launch_bounds__(THREADS)
kmers_tile_kernel(StreamView s, int k, int canonical, uint64_t seed, uint64_t *__restrict__ out_lo,
                  uint64_t *__restrict__ out_hi, uint64_t *__restrict__ out_hash, uint64_t *__restrict__ out_not,
                  uint32_t n_tiles, Counters *ctr)
{
    constexpr int PER_THREAD = WIDE ? 8 : 16;
    constexpr int TILE = THREADS * PER_THREAD, WAVE_POS = 64 * PER_THREAD, PITCH = PER_THREAD + 1;
    __shared__ uint64_t xp[THREADS / 64][64 * PITCH];
    const int tid = (int)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    uint64_t *X = xp[wv];
    const uint64_t last_b = s.n_bwords - 1, last_m = s.n_mwords - 1;
    const int sk = (WIDE ? 128 : 64) - 2 * k;
    const uint64_t kmask_shift = 64 - k;
    const uint64_t canon_mask = canonical ? ~0ULL : 0ULL;
    uint64_t n_valid = 0;
#pragma unroll 1
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t p0 = (uint64_t)tile * TILE + (uint64_t)tid * PER_THREAD;
        const uint64_t bw = p0 >> 5, mw = p0 >> 6;
        const uint64_t w0 = s.bases[bw < last_b ? bw : last_b];
        const uint64_t w1 = s.bases[bw + 1 < last_b ? bw + 1 : last_b];
        const uint64_t w2 = WIDE ? s.bases[bw + 2 < last_b ? bw + 2 : last_b] : 0;
        const uint64_t m0 = s.mask[mw < last_m ? mw : last_m];
        const uint64_t m1 = s.mask[mw + 1 < last_m ? mw + 1 : last_m];
        const int o = 2 * (int)(p0 & 31);
        const uint64_t v0 = o ? (w0 << o) | (w1 >> (64 - o)) : w0;
        const uint64_t v1 = o ? (w1 << o) | (WIDE ? w2 >> (64 - o) : 0) : w1;
        const uint64_t v2 = WIDE ? (o ? w2 << o : w2) : 0;
        const int ms = (int)(p0 & 63);
        const uint64_t mh = ms ? (m0 << ms) | (m1 >> (64 - ms)) : m0;
        const uint64_t ml = WIDE ? (ms ? m1 << ms : m1) : 0;
        uint64_t klo[PER_THREAD], khi[WIDE ? PER_THREAD : 1];
        uint32_t notbits = 0;                                // bit (PER_THREAD - 1 - j): no k-mer at p0 + j
        uint64_t rch = 0, rcl = 0;
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            uint64_t kh = 0, kl;
            bool bad;
            if (!WIDE) {
                const uint32_t a2 = (uint32_t)(v0 >> 32), a1 = (uint32_t)v0, a0 = (uint32_t)(v1 >> 32);      // (as in scan_part)
                const uint64_t win = j ? ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, 32 - 2 * j) << 32) | __builtin_amdgcn_alignbit(a1, a0, 32 - 2 * j) : v0;
                const uint64_t fwd = win >> sk;
                if (j == 0) rcl = (~rev_pairs64(fwd)) >> sk;
                else rcl = (rcl >> 2) | ((uint64_t)(3u - (uint32_t)(fwd & 3)) << (2 * k - 2));
                bad = ((mh << j) >> kmask_shift) != 0;
                kl = __builtin_amdgcn_inverse_ballot_w64(__builtin_amdgcn_ballot_w64(rcl < fwd) & canon_mask) ? rcl : fwd;
            } else {
                const uint64_t A = j ? (v0 << (2 * j)) | (v1 >> (64 - 2 * j)) : v0;
                const uint64_t B = j ? (v1 << (2 * j)) | (v2 >> (64 - 2 * j)) : v1;
                const uint64_t fh = sk ? A >> sk : A;
                const uint64_t fl = sk ? (B >> sk) | (A << (64 - sk)) : B;
                if (j == 0) {
                    const uint64_t th = ~rev_pairs64(fl), tl = ~rev_pairs64(fh);
                    rch = sk ? th >> sk : th;
                    rcl = sk ? (tl >> sk) | (th << (64 - sk)) : tl;
                } else {
                    rcl = (rcl >> 2) | (rch << 62);
                    rch = (rch >> 2) | ((uint64_t)(3u - (uint32_t)(fl & 3)) << (2 * k - 2 - 64));
                }
                const uint64_t mx = j ? (mh << j) | (ml >> (64 - j)) : mh;
                bad = (mx >> kmask_shift) != 0;
                const bool use_rc = canonical && (rch < fh || (rch == fh && rcl < fl));
                kh = use_rc ? rch : fh;
                kl = use_rc ? rcl : fl;
            }
            const bool valid = !bad && p0 + j < s.n_bases;
            klo[j] = valid ? kl : 0;
            if constexpr (WIDE) khi[j] = valid ? kh : 0;
            notbits |= (valid ? 0u : 1u) << (PER_THREAD - 1 - j);
            n_valid += valid;
        }
        // the wave's positions [wave0, wave0 + WAVE_POS): store instruction i writes positions wave0 + 64 i + lane
        const uint64_t wave0 = (uint64_t)tile * TILE + (uint64_t)wv * WAVE_POS;
        auto emit = [&](const uint64_t (&vals)[PER_THREAD], uint64_t *__restrict__ dst) {
#pragma unroll
            for (int j = 0; j < PER_THREAD; j++) X[lane * PITCH + j] = vals[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < PER_THREAD; i++) {
                const int e = i * 64 + lane;
                const uint64_t v = X[(e / PER_THREAD) * PITCH + (e % PER_THREAD)];
                const uint64_t p = wave0 + (uint64_t)e;
                if (p < s.n_bases) dst[p] = v;
            }
            __builtin_amdgcn_wave_barrier();
        };
        emit(klo, out_lo);
        if constexpr (WIDE) {
            if (out_hi) emit(khi, out_hi);
        }
        if (out_hash) {
            uint64_t hs[PER_THREAD];
#pragma unroll
            for (int j = 0; j < PER_THREAD; j++) {
                const bool valid = !((notbits >> (PER_THREAD - 1 - j)) & 1u);
                hs[j] = valid ? fmix64(klo[j] ^ hash_tweak<WIDE>(WIDE ? khi[j] : 0, seed)) : 0;
            }
            emit(hs, out_hash);
        }
        if (out_not) {
            // 64 / PER_THREAD lanes make one mask word (MSB first)
            constexpr int LPW = 64 / PER_THREAD;
            uint64_t word = (uint64_t)notbits << (64 - PER_THREAD * (1 + (lane % LPW)));
#pragma unroll
            for (int d = 1; d < LPW; d <<= 1) word |= (uint64_t)__shfl_xor((unsigned long long)word, d);
            const uint64_t pw = wave0 + (uint64_t)(lane / LPW) * 64;
            if ((lane % LPW) == 0 && pw < ((s.n_bases + 63) & ~63ULL)) out_not[pw >> 6] = word;
        }
    }
    n_valid = wave_sum(n_valid);
    if (lane == 0 && n_valid) atomicAdd(&ctr->n_valid, (unsigned long long)n_valid);
}

// ---- level 2: one workgroup per tile of a level-1 piece; records go to per-SEGMENT regions through
// global cursors.  Shared write frontiers keep the DRAM pages and L2 lines being written few and
// hot (every resident workgroup appends to the same 2^b2 segments of one coarse bin at a time),
// which measured faster than private level-2 pieces; the cursor atomics are issued before the
// scatter phase and only waited for after it, so their latency is covered.
template <int THREADS, int PER_THREAD, int MIN_WAVES, class R>
__global__ void __launch_bounds__(THREADS, MIN_WAVES)
repart_kernel(const R *__restrict__ in, const uint32_t *__restrict__ cnt1, uint32_t G, uint32_t capw,
              uint32_t tiles_per_piece, int b1, int b2, uint32_t cap2, R *__restrict__ out,
              uint32_t *__restrict__ cursor2, OvfList<R> ovf, Counters *ctr, int xcd_affine = 0, uint32_t bin_skew = 0)
{
    constexpr int TILE = THREADS * PER_THREAD;
    constexpr int NB = THREADS >= MAX_BINS2 ? MAX_BINS2 : MAX_BINS;      // the bin scan is one thread per bin
    __shared__ SplitLds<THREADS, PER_THREAD, R, NB, false> L;
    const int tid = (int)threadIdx.x;
    // one-dimensional grid (the number of bins can exceed the 65535 of grid.y): bin-major, then piece, then tile
    const uint32_t per_bin = G * tiles_per_piece;
    uint32_t b = blockIdx.x / per_bin, bx = blockIdx.x % per_bin;
    if (xcd_affine) {
        // eight bins at a time, one per XCD (consecutive blocks are dealt round-robin over the XCDs): all tiles of a bin
        // then append to its 2^b2 frontiers through ONE L2, which assembles whole lines (speed only)
        const uint32_t slot = blockIdx.x >> 3;
        b = 8 * (slot / per_bin) + (blockIdx.x & 7);
        bx = slot % per_bin;
    }
    const uint32_t w = bx / tiles_per_piece, t0 = (bx % tiles_per_piece) * TILE;
    const uint64_t piece = (uint64_t)b * G + w;
    uint32_t n = cnt1[piece];
    if (n > capw) n = capw;
    if (t0 >= n) return;
    const int nbins = 1 << b2;
    const int shift = 64 - b1 - b2;
    auto bin_of = [=](uint64_t h) -> uint32_t { return (uint32_t)(h >> shift) & (uint32_t)(nbins - 1); };
    for (int i = tid; i < NB; i += THREADS) L.cnt[i] = 0;
    if (tid == 0) L.ovf_seen = 0;
    const R *src = in + piece * capw + (uint64_t)b * bin_skew;
    R hs[PER_THREAD];
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++) {
        const uint32_t i = t0 + (uint32_t)j * THREADS + tid;
        hs[j] = src[i < n ? i : 0];
    }
    __syncthreads();
    uint32_t valid = 0;
    uint32_t rk[PER_THREAD];
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++) {
        const uint32_t i = t0 + (uint32_t)j * THREADS + tid;
        rk[j] = 0;
        if (i < n) {
            valid |= 1u << j;
            rk[j] = atomicAdd(&L.cnt[bin_of(hs[j].h)], 1u);
        }
    }
    lds_barrier();                                       // A
    // scan (lane = bin) and reserve the segment ranges; the atomics' results are used after the scatter
    uint32_t *cursor = cursor2 + ((uint64_t)b << b2);
    const int wv = tid >> 6, lane = tid & 63;
    uint32_t g = 0, ex = 0, c_mine = 0;
    if (wv * 64 < nbins) {
        const uint32_t c = tid < nbins ? L.cnt[tid] : 0;
        c_mine = c;
        uint32_t below = 0;
#pragma unroll
        for (int v = 0; v < NB / 64 - 1; v++) {
            const uint32_t x = L.cnt[64 * v + lane];
            below += v < wv ? x : 0u;
        }
        below = wave_total(below);
        ex = below + wave_incl_scan(c) - c;
        if (tid < nbins) {
            L.off[tid] = ex;
            if (c) g = atomicAdd(&cursor[tid], c);
            if (tid == nbins - 1) L.total = ex + c;
        }
    }
    lds_barrier();                                       // B
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++)
        if ((valid >> j) & 1u) L.stage[L.off[bin_of(hs[j].h)] + rk[j]] = hs[j];
    if (tid < nbins) {
        L.delta[tid] = g - ex;
        if (g + c_mine > cap2) L.ovf_seen = 1;
    }
    lds_barrier();                                       // C
    const uint32_t total = L.total;
    const uint64_t seg0 = (uint64_t)b << b2;
    uint32_t n_overflow = 0;
    if (!L.ovf_seen) {
        // every segment region still has room for this tile: no bounds check, no overflow ballot
#pragma unroll 4
        for (uint32_t i = tid; i < total; i += THREADS) {
            const R rec = L.stage[i];
            const uint32_t bin = bin_of(rec.h);
            const uint32_t idx = i + L.delta[bin];       // 32-bit on purpose: delta is a wrapped difference
            out[(seg0 + bin) * cap2 + idx] = rec;
        }
    } else {
#pragma unroll 2
        for (uint32_t i = tid; i < total; i += THREADS) {
            const R rec = L.stage[i];
            const uint32_t bin = bin_of(rec.h);
            const uint32_t idx = i + L.delta[bin];
            if (idx < cap2) out[(seg0 + bin) * cap2 + idx] = rec;
            ovf_append(ovf, idx >= cap2, rec, n_overflow);
        }
    }
    n_overflow = (uint32_t)wave_sum(n_overflow);
    if (lane_id() == 0 && n_overflow) atomicAdd(&ctr->n_overflow, (unsigned long long)n_overflow);
}

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
    for (int q = 0; q < SEG_VEC; q++) r.v[q] = src[q * SEG_THREADS + (int)threadIdx.x];
    return r;
}

__device__ __forceinline__ void stage_segment(uint32_t *seg, const SegRegs &r)
{
    uint4 *dst = (uint4 *)seg;
#pragma unroll
    for (int q = 0; q < SEG_VEC; q++) dst[q * SEG_THREADS + (int)threadIdx.x] = r.v[q];
}

template <class R>
DK_SEG_KERNEL
seg_insert_kernel(unsigned long long *filter, PieceList<R> pl, int n_hashes, int blk_shift)
{
    __shared__ __attribute__((aligned(16))) uint32_t seg[SEG_WORDS32];
    const uint64_t seg_id = segment_of_block();
    const SegCounts sc = seg_counts(pl, seg_id >> pl.sbits);
    const SegRegs sr = fetch_segment(filter, seg_id);
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) return;                       // nothing to add: leave the segment untouched
    constexpr int UNROLL = 8;
    uint64_t h[UNROLL];
    bool have[UNROLL];
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            h[u] = sp.at(have[u] ? i : 0).h;
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
    uint4 *dst = (uint4 *)filter + seg_id * (SEG_BYTES / 16);
    const uint4 *s4 = (const uint4 *)seg;
    for (int i = (int)threadIdx.x; i < SEG_BYTES / 16; i += SEG_THREADS) dst[i] = s4[i];
}

// Where the absent records of a segment go.
//   per batch (ACC = false): recs[seg * cap ...], compacted per wave by ballot; cnt[seg] = their number
//   accumulate (ACC = true, dk_accum_add): the accumulator's counting units of the segment -- unit = seg << sub_bits |
//     the next sub_bits hash bits -- appended behind cnt[unit], which persists from batch to batch; a record whose unit
//     is full goes to the accumulator's overflow list
// seg = the segment's index inside the window (= blockIdx.x); the filter is addressed with seg_base + seg.
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

template <class R, bool ACC>
struct MissSink {
    uint32_t *sfill;       // LDS: ACC: fill of the segment's units; else [0] = absent records so far
    R *dst;
    const MissOut<R> &mo;
    uint64_t seg;
    uint32_t n_dropped = 0;
    __device__ __forceinline__ MissSink(uint32_t *lds, const MissOut<R> &m, uint64_t seg_local) : sfill(lds), mo(m), seg(seg_local)
    {
        if constexpr (ACC) {
            dst = m.recs + (seg_local << m.sub_bits) * (uint64_t)m.cap;
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
        if constexpr (ACC) {
            uint32_t sub = 0, pos = 0;
            if (absent) {
                sub = (uint32_t)(rec.h >> mo.sub_shift) & ((1u << mo.sub_bits) - 1u);
                pos = atomicAdd(&sfill[sub], 1u);
            }
            const bool full = absent && pos >= mo.cap;
            if (absent && !full) dst[(uint64_t)sub * mo.cap + pos] = rec;
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
        if constexpr (ACC) {
            const uint32_t ws = wave_total(my_absent);
            if (lane_id() == 0 && ws) atomicAdd(&sfill[SUB_TALLY], ws);
            lds_barrier();
            if (threadIdx.x < (1u << mo.sub_bits)) {
                const uint32_t f = sfill[threadIdx.x];
                mo.cnt[(seg << mo.sub_bits) + threadIdx.x] = f < mo.cap ? f : mo.cap;
            }
            if (threadIdx.x == 0 && sfill[SUB_TALLY]) atomicAdd(&ctr->shard[blockIdx.x % COUNTER_SHARDS], (unsigned long long)sfill[SUB_TALLY]);
            n_dropped = (uint32_t)wave_sum(n_dropped);
            if (lane_id() == 0 && n_dropped) atomicAdd(&ctr->n_overflow, (unsigned long long)n_dropped);
        } else {
            if (threadIdx.x == 0) {
                mo.cnt[seg] = sfill[0];
                if (sfill[0]) atomicAdd(&ctr->shard[blockIdx.x % COUNTER_SHARDS], (unsigned long long)sfill[0]);
            }
        }
    }
};

// NH > 0: the number of hash bits is a compile-time constant (the four LDS reads of a record are then
// issued back to back instead of one by one behind the short-circuit test); NH == 0: n_hashes at run time
template <class R, int NH, bool ACC>
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
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) {
        if (!ACC && threadIdx.x == 0) mo.cnt[seg_id] = 0;
        return;
    }
    constexpr int UNROLL = 8;                 // records in flight per thread: loads first, then the LDS tests
    R rec[UNROLL];
    bool have[UNROLL];
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            rec[u] = sp.at(have[u] ? i : 0);
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
template <class R>
DK_SEG_KERNEL
seg_exact_insert_kernel(unsigned long long *table, PieceList<R> pl, int T, Counters *ctr)
{
    constexpr bool WIDE = sizeof(R) == 16;
    static_assert(SEG_BYTES == EXACT_SEG_WORDS * 8, "exact segments are the filter segments");
    __shared__ __attribute__((aligned(16))) unsigned long long tab[EXACT_SEG_WORDS];
    const uint64_t seg_id = segment_of_block();
    const SegCounts sc = seg_counts(pl, seg_id >> pl.sbits);
    const SegRegs sr = fetch_segment(table, seg_id);
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) return;
    constexpr int UNROLL = 8;
    R rec[UNROLL];
    bool have[UNROLL];
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            rec[u] = sp.at(have[u] ? i : 0);
        }
    };
    fetch(0);
    stage_segment((uint32_t *)tab, sr);
    __syncthreads();
    const uint64_t EMPTY = exact_empty(seg_id, T);
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
    uint4 *dst = (uint4 *)table + seg_id * (SEG_BYTES / 16);
    const uint4 *s4 = (const uint4 *)tab;
    for (int i = (int)threadIdx.x; i < SEG_BYTES / 16; i += SEG_THREADS) dst[i] = s4[i];
    n_full = (uint32_t)wave_sum(n_full);
    if (lane_id() == 0 && n_full) atomicAdd(&ctr->n_set_full, (unsigned long long)n_full);
}

// One workgroup per segment, eight records in flight per thread, 32-byte bucket probes.  (A persistent
// walk over the segments with the table and the records fetched in one round trip measured 6.8 ms
// against 5.1 ms for this form at 2^17 segments: the hardware's workgroup scheduler overlaps the
// segments' load / probe phases better than two resident persistent workgroups per CU do.)
template <class R, bool ACC>
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
    const SegPieces<R> sp = seg_pieces(pl, seg_id >> pl.sbits, sc);
    const uint32_t n = sp.total();
    if (n == 0) {
        if (!ACC && threadIdx.x == 0) mo.cnt[seg_id] = 0;
        return;
    }
    constexpr int UNROLL = 8;
    R rec[UNROLL];
    bool have[UNROLL];
    auto fetch = [&](uint32_t i0) {
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const uint32_t i = i0 + (uint32_t)u * SEG_THREADS + threadIdx.x;
            have[u] = i < n;
            rec[u] = sp.at(have[u] ? i : 0);
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

// Exact counting of one segment's absent records.
// Most absent k-mers are singletons (sequencing errors), so a hash table for all of them is wasted
// work.  Two 64-Kbit LDS bitmaps classify the records first: bit(h) set twice => the record MAY have
// a twin (true duplicate or bitmap collision) and goes through a small LDS hash table; every other
// record is provably unique and is emitted with count 1 straight from registers.  Exact for any
// input: all copies of a k-mer share a bit, so all of them are flagged.
// Two geometries: <512 threads, 2048 slots, 64-Kbit bitmaps> for segments with thousands of absent
// records, <128, 512, 8 Kbit> (8 KB of LDS, many workgroups per CU) when a segment holds a few hundred.
template <int CNT_THREADS, int CNT_SLOTS, int CNT_BM_WORDS, bool WIDE>
__global__ void __launch_bounds__(CNT_THREADS) __attribute__((amdgpu_waves_per_eu(4)))
seg_count_kernel(PieceList<typename RecOf<WIDE>::type> pl, uint64_t n_seg, int T, uint64_t seed, uint32_t min_count,
                 uint64_t region_cap, uint64_t *__restrict__ out_kmer, uint64_t *__restrict__ out_hi,
                 uint32_t *__restrict__ out_cnt, Counters *ctr, uint64_t unit_base)
{
    // unit_base: the units counted are unit_base .. unit_base + n_seg of the 2^T hash-prefix ranges (a window of
    // an accumulator).  Entries beyond a region's capacity are not written but still tallied in region_fill (and
    // flagged through n_overflow), so a table sized too small tells how large it has to be.
    // Every barrier of this kernel orders LDS traffic only (lds_barrier): records are read-only, results write-only, and
    // a __syncthreads() would make every unit wait until the result stores of the previous phase have left the CU.
    using R = typename RecOf<WIDE>::type;
    constexpr int CNT_RPT = (WIDE && CNT_THREADS < 1024) ? 8 : 16;   // records held per thread (the 1024-thread geometry runs one workgroup per CU: 128 VGPRs)
    constexpr int CNT_CHUNK = CNT_THREADS * CNT_RPT;
    // k <= 32: the table key is the record's hash (EMPTY = a value outside this segment's prefix).
    // k > 32: the key is a 64-bit fingerprint of (h, hi) (EMPTY = 0); the slot's owner stores (h, hi)
    // beside it and every record re-checks the full key after the insert phase, so a fingerprint
    // collision is detected (and the batch redone exactly) instead of merging two k-mers.
    __shared__ unsigned long long keys[CNT_SLOTS];
    __shared__ unsigned long long key_h[WIDE ? CNT_SLOTS : 1], key_hi[WIDE ? CNT_SLOTS : 1];
    __shared__ uint32_t cnts[CNT_SLOTS];
    __shared__ uint32_t bm_a[CNT_BM_WORDS], bm_b[CNT_BM_WORDS];
    __shared__ uint32_t wave_sums[CNT_THREADS / 64];
    __shared__ uint32_t total;
    __shared__ unsigned long long gbase;
    constexpr bool GATHER = !(WIDE && CNT_THREADS == 1024);     // (that geometry has no registers to spare)
    constexpr uint32_t WB = WIDE ? 32 : 64;              // per wave: flagged records gathered for one dense trip through the table
    __shared__ R wbuf[GATHER ? CNT_THREADS / 64 : 1][WB];
    const int tid = (int)threadIdx.x;
    const int wave = tid >> 6;
    uint32_t n_distinct = 0, n_fail = 0;
    // this workgroup appends to output region `region` through that region's own fill counter
    const uint32_t region = blockIdx.x % RESULT_REGIONS;
    unsigned long long *fill = &ctr->region_fill[region];
    const uint64_t region_base = (uint64_t)region * region_cap;
    auto fp_of = [](const R &rec) -> unsigned long long {
        if constexpr (WIDE) {
            const unsigned long long f = fmix64(rec.h ^ (rec_hi(rec) * 0x9E3779B97F4A7C15ULL));
            return f ? f : 1ULL;
        } else {
            return rec.h;
        }
    };
    // persistent: a workgroup walks segments blockIdx.x, +gridDim.x, ... (launching one tiny
    // workgroup per segment cost ~50 ns of wall time each at 2^18 segments)
    for (uint64_t seg_id = blockIdx.x; seg_id < n_seg; seg_id += gridDim.x) {
        const SegPieces<R> sp = seg_pieces(pl, seg_id);
        const uint32_t n = sp.total();
        if (n == 0) continue;
        const unsigned long long EMPTY = WIDE ? 0ULL : (unsigned long long)((seg_id + unit_base) ^ 1ULL) << (64 - T);
        const uint32_t n_chunks = (n + CNT_CHUNK - 1) / CNT_CHUNK;
        const bool single = n_chunks == 1;                 // the common case: the records stay in registers
        R hv[CNT_RPT];
        auto load_chunk = [&](uint32_t c) {
#pragma unroll
            for (int u = 0; u < CNT_RPT; u++) {
                const uint32_t i = c * CNT_CHUNK + (uint32_t)u * CNT_THREADS + tid;
                hv[u] = sp.at(i < n ? i : 0);
            }
        };
        auto have = [&](uint32_t c, int u) -> bool { return c * CNT_CHUNK + (uint32_t)u * CNT_THREADS + tid < n; };
        // bitmap of >= 16 bits per record where the geometry has them (<= 6 % of the unique records collide and take the
        // table path), a power of two up to CNT_BM_WORDS
        uint32_t bm_words = 64;
        while (bm_words * 2 < n && bm_words < (uint32_t)CNT_BM_WORDS) bm_words <<= 1;
        const uint32_t bm_mask = bm_words * 32 - 1;
        auto bit_of = [=](const R &rec, uint32_t &w, uint32_t &m) {
            const uint32_t b = (uint32_t)(fp_of(rec) >> 20) & bm_mask;
            w = b >> 5;
            m = 1u << (b & 31);
        };
        // k > 32: the bitmap position of a record costs an fmix64; with the records in registers (single) it is
        // computed once and kept, like the verdict of pass 2 (fbits) that passes 3 and 4 would otherwise re-derive
        constexpr bool KEEP_BITS = WIDE && CNT_RPT == 8;
        uint32_t bidx[KEEP_BITS ? CNT_RPT : 1];
        uint32_t fbits = 0;                                // bit u: record u of this thread may have a twin (single only)
        for (uint32_t i = tid; i < bm_words; i += CNT_THREADS) { bm_a[i] = 0; bm_b[i] = 0; }
        if (single) load_chunk(0);
        lds_barrier();
        // pass 1: mark
        for (uint32_t c = 0; c < n_chunks; c++) {
            if (!single) load_chunk(c);
#pragma unroll
            for (int u = 0; u < CNT_RPT; u++) {
                if (!have(c, u)) continue;
                uint32_t w, m;
                bit_of(hv[u], w, m);
                if constexpr (KEEP_BITS) bidx[u] = (w << 5) | (uint32_t)__builtin_ctz(m);
                if (atomicOr(&bm_a[w], m) & m) atomicOr(&bm_b[w], m);
            }
        }
        lds_barrier();
        // pass 2: classify; per-wave count of provably unique records, block count of flagged ones
        uint32_t my_unique = 0, my_flagged = 0;
        for (uint32_t c = 0; c < n_chunks; c++) {
            if (!single) load_chunk(c);
#pragma unroll
            for (int u = 0; u < CNT_RPT; u++) {
                if (!have(c, u)) continue;
                uint32_t w, m;
                if (KEEP_BITS && single) { w = bidx[u] >> 5; m = 1u << (bidx[u] & 31); }
                else bit_of(hv[u], w, m);
                const bool fl = (bm_b[w] & m) != 0;
                if (fl) my_flagged++; else my_unique++;
                if constexpr (KEEP_BITS) fbits |= (fl ? 1u : 0u) << u;
            }
        }
        auto flagged = [&](int u, const R &rec) -> bool {
            if constexpr (KEEP_BITS) {
                if (single) return (fbits >> u) & 1u;
            }
            uint32_t w, m;
            bit_of(rec, w, m);
            return (bm_b[w] & m) != 0;
        };
        const uint32_t emit_unique = min_count <= 1 ? 1u : 0u;
        const uint32_t wave_unique = wave_total(my_unique);          // uniform per wave
        (void)block_excl_scan<true>(my_flagged, wave_sums, &total);
        const uint32_t n_flagged = total;
        lds_barrier();
        // bases of the waves' unique runs: prefix over the per-wave totals
        if ((tid & 63) == 0) wave_sums[wave] = wave_unique;
        lds_barrier();
        uint32_t wave_base = 0, all_unique = 0;
#pragma unroll
        for (int v = 0; v < CNT_THREADS / 64; v++) {
            const uint32_t x = wave_sums[v];
            if (v < wave) wave_base += x;
            all_unique += x;
        }
        n_distinct += (tid == 0) ? all_unique : 0;
        if (tid == 0) gbase = (emit_unique && all_unique) ? atomicAdd(fill, (unsigned long long)all_unique) : 0ULL;
        lds_barrier();
        // pass 3: emit the unique records, each wave a contiguous run, compacted by ballot
        if (emit_unique && all_unique) {
            uint64_t o = gbase + wave_base;
            for (uint32_t c = 0; c < n_chunks; c++) {
                if (!single) load_chunk(c);
#pragma unroll
                for (int u = 0; u < CNT_RPT; u++) {
                    const bool uniq = have(c, u) && !flagged(u, hv[u]);
                    const uint64_t bal = __ballot(uniq);
                    if (uniq) {
                        const uint64_t pos = o + (uint64_t)popc_below(bal);
                        if (pos < region_cap) {
                            out_kmer[region_base + pos] = rec_lo(hv[u], seed);
                            if constexpr (WIDE) out_hi[region_base + pos] = rec_hi(hv[u]);
                            out_cnt[region_base + pos] = 1;
                        } else {
                            n_fail++;               // region full: host redoes the batch with the direct family
                        }
                    }
                    o += (uint64_t)__popcll(bal);
                }
            }
        }
        // Flagged records: exact counts in the LDS hash table, one sub-range of the key space per
        // round.  The number of rounds starts from a guess (8 copies per key) and a round whose keys
        // do not fit is split in four and redone -- nothing of it has been emitted yet -- so a segment
        // holding a million copies of one k-mer costs one round, not a thousand.
        if (n_flagged) {
            uint32_t slots = 256;
            while (slots < 2 * n_flagged && slots < (uint32_t)CNT_SLOTS) slots <<= 1;
            const uint32_t slot_mask = slots - 1;
            uint32_t rounds = (n_flagged + 4 * slots - 1) / (4 * slots);
            uint32_t r = 0;
            while (r < rounds) {
                lds_barrier();
                for (uint32_t i = tid; i < slots; i += CNT_THREADS) { keys[i] = EMPTY; cnts[i] = 0; }
                if (tid == 0) total = 0;                       // doubles as the "round does not fit" flag
                lds_barrier();
                for (int phase = 0; phase < (WIDE ? 2 : 1); phase++) {
                    // phase 0: insert and count.  phase 1 (k > 32): re-check the full key of every record
                    if (GATHER && single) {
                        // Records in registers.  Flagged records are few per register slot (~10 % of the lanes when nearly every
                        // k-mer is unique), and a trip through the table is a chain of LDS round trips: going slot by slot
                        // cost eight (sixteen) chains per wave with a handful of lanes each -- half of the kernel's time.  The
                        // wave gathers its flagged records in a 64-entry LDS buffer instead and walks the table with all
                        // lanes busy, once per 64 records.
                        R *const wb = wbuf[GATHER ? wave : 0];
                        auto walk = [&](uint32_t cnt) {
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            const uint32_t lane = (uint32_t)lane_id();
                            bool active = lane < cnt;
                            const R cur = wb[lane & (WB - 1)];
                            const unsigned long long f = fp_of(cur);
                            uint32_t slot = (uint32_t)(f >> 8) & slot_mask, tries = 0;
                            while (__any(active)) {
                                if (active) {
                                    if (phase == 0) {
                                        const unsigned long long prev = atomicCAS(&keys[slot], EMPTY, f);
                                        if (prev == EMPTY) {
                                            if constexpr (WIDE) { key_h[slot] = cur.h; key_hi[slot] = rec_hi(cur); }
                                        }
                                        if (prev == EMPTY || prev == f) {
                                            atomicAdd(&cnts[slot], 1u);
                                            active = false;
                                        } else {
                                            slot = (slot + 1) & slot_mask;
                                            if (++tries == 64) { total = 1; active = false; }      // too crowded: split this round
                                        }
                                    } else {
                                        if (keys[slot] == f) {
                                            if (key_h[slot] != cur.h || key_hi[slot] != rec_hi(cur)) n_fail++;
                                            active = false;
                                        } else {
                                            slot = (slot + 1) & slot_mask;
                                            if (++tries == 64) { n_fail++; active = false; }
                                        }
                                    }
                                }
                            }
                            __builtin_amdgcn_wave_barrier();
                        };
                        uint32_t held = 0;                                 // records in the wave's buffer (wave-uniform)
#pragma unroll
                        for (int u = 0; u < CNT_RPT; u++) {
                            bool want = have(0, u) && flagged(u, hv[u]);
                            if (want && rounds > 1) {
                                const unsigned long long fu = fp_of(hv[u]);
                                want = (uint32_t)((((fu >> 36) & 0xFFFFF) * (uint64_t)rounds) >> 20) == r;
                            }
#pragma unroll
                            for (uint32_t part = 0; part < 64 / WB; part++) {          // at most WB records join at a time
                                const bool mine = want && (WB == 64 || (uint32_t)lane_id() / WB == part);
                                const uint64_t m = __ballot(mine);
                                const uint32_t c = (uint32_t)__popcll(m);
                                if (c == 0) continue;
                                if (held + c > WB) { walk(held); held = 0; }
                                if (mine) wb[held + (uint32_t)popc_below(m)] = hv[u];
                                held += c;
                            }
                        }
                        if (held) walk(held);
                    } else
                    for (uint32_t c = 0; c < n_chunks; c++) {
                        if (!single) load_chunk(c);
#pragma unroll
                        for (int u = 0; u < CNT_RPT; u++) {
                            if (!have(c, u)) continue;
                            const R rec = hv[u];
                            if (!flagged(u, rec)) continue;
                            const unsigned long long f = fp_of(rec);
                            const uint32_t rr = (uint32_t)((((f >> 36) & 0xFFFFF) * (uint64_t)rounds) >> 20);
                            if (rr != r) continue;
                            uint32_t slot = (uint32_t)(f >> 8) & slot_mask;
                            uint32_t tries = 0;
                            if (phase == 0) {
                                for (; tries < 64; tries++) {
                                    const unsigned long long prev = atomicCAS(&keys[slot], EMPTY, f);
                                    if (prev == EMPTY) {
                                        if constexpr (WIDE) { key_h[slot] = rec.h; key_hi[slot] = rec_hi(rec); }
                                    }
                                    if (prev == EMPTY || prev == f) { atomicAdd(&cnts[slot], 1u); break; }
                                    slot = (slot + 1) & slot_mask;
                                }
                                if (tries == 64) total = 1;        // too crowded: split this round
                            } else {
                                for (; tries < 64 && keys[slot] != f; tries++) slot = (slot + 1) & slot_mask;
                                if (tries == 64 || key_h[slot] != rec.h || key_hi[slot] != rec_hi(rec)) n_fail++;
                            }
                        }
                    }
                    lds_barrier();
                    if (total) break;
                }
                if (total) {
                    if (rounds >= (1u << 18)) { n_fail++; r = rounds; break; }   // cannot split further: redo on the direct family
                    rounds *= 4;
                    r *= 4;
                    continue;
                }
                uint32_t mine = 0;
                for (uint32_t sl = tid; sl < slots; sl += CNT_THREADS)
                    if (keys[sl] != EMPTY) { n_distinct++; if (cnts[sl] >= min_count) mine++; }
                const uint32_t ex = block_excl_scan<true>(mine, wave_sums, &total);
                if (tid == 0) gbase = total ? atomicAdd(fill, (unsigned long long)total) : 0ULL;
                lds_barrier();
                uint64_t o = gbase + ex;
                for (uint32_t sl = tid; sl < slots; sl += CNT_THREADS) {
                    if (keys[sl] != EMPTY && cnts[sl] >= min_count) {
                        if (o < region_cap) {
                            if constexpr (WIDE) {
                                const Rec2 kr{key_h[sl], key_hi[sl]};
                                out_kmer[region_base + o] = rec_lo(kr, seed);
                                out_hi[region_base + o] = kr.hi;
                            } else {
                                out_kmer[region_base + o] = unfmix64(keys[sl]) ^ seed;
                            }
                            out_cnt[region_base + o] = cnts[sl];
                        } else {
                            n_fail++;
                        }
                        o++;
                    }
                }
                r++;
            }
        }
        lds_barrier();
    }
    n_distinct = (uint32_t)wave_sum(n_distinct);
    n_fail = (uint32_t)wave_sum(n_fail);
    if (lane_id() == 0) {
        if (n_distinct) atomicAdd(&ctr->n_distinct, (unsigned long long)n_distinct);
        if (n_fail) atomicAdd(&ctr->n_overflow, (unsigned long long)n_fail);
    }
}

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
// appended to `miss` and tallied per segment for the CSR build.
template <class R>
__global__ void __launch_bounds__(DIRECT_BLOCK)
ovf_probe_kernel(unsigned long long *filter, OvfList<R> ovf, int log2_blocks, int n_hashes, int exact_T, int T,
                 uint64_t unit_base, R *__restrict__ miss, uint32_t *seg_hist, Counters *ctr)
{
    unsigned long long n = *ovf.count;
    if (n > ovf.cap) n = ovf.cap;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (n + 63) & ~63ULL;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool have = i < n;
        const R rec = ovf.recs[have ? i : 0];
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
template <class R>
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
            if (!full) mo.recs[unit * mo.cap + pos] = rec;
        }
        if (__ballot(full)) ovf_append(mo.ovf, full, rec, n_dropped);
    }
    n_dropped = (uint32_t)wave_sum(n_dropped);
    if (lane_id() == 0 && n_dropped) atomicAdd(&ctr->n_overflow, (unsigned long long)n_dropped);
}

// the same for k-mers (the candidate list of the direct family: the exact redo path of a batch whose partition
// overflowed); k-mers outside the window are skipped; the appended ones are tallied in Counters::shard
template <bool WIDE>
__global__ void __launch_bounds__(DIRECT_BLOCK)
acc_append_kmers_kernel(const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi, uint64_t n, uint64_t seed,
                        int wbits, uint32_t widx, int T, uint64_t unit_base, MissOut<typename RecOf<WIDE>::type> mo, Counters *ctr)
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
        bool full = false;
        if (have) {
            n_in++;
            const uint64_t unit = (rec.h >> (64 - T)) - unit_base;
            const uint32_t pos = atomicAdd(&mo.cnt[unit], 1u);
            full = pos >= mo.cap;
            if (!full) mo.recs[unit * mo.cap + pos] = rec;
        }
        if (__ballot(full)) ovf_append(mo.ovf, full, rec, n_dropped);
    }
    n_dropped = (uint32_t)wave_sum(n_dropped);
    n_in = wave_sum(n_in);
    if (lane_id() == 0) {
        if (n_dropped) atomicAdd(&ctr->n_overflow, (unsigned long long)n_dropped);
        if (n_in) atomicAdd(&ctr->shard[blockIdx.x % COUNTER_SHARDS], (unsigned long long)n_in);
    }
}

// ---- host side -----------------------------------------------------------------------------------------
// Piece capacity = mean + 8 sigma.  Records per piece are not Poisson: a k-mer seen m times
// (coverage, repeats) lands all its copies in one segment, so the variance is mean * ratio with
// ratio = E[m^2]/E[m].  Level-1 pieces see 1/G of the reads, so copies of one k-mer rarely meet
// there (ratio 4 allowed); level-2 pieces gather a whole segment (ratio 64 allowed, 30-60x
// coverage; dk_engine_set_option("multiplicity_hint") lowers it for batches that cover the genome
// only a few times, e.g. one of the ~40 batches of a 30x whole-genome sample).  Anything heavier
// (poly-A style heavy hitters) goes to the overflow list, and if that overflows too the batch
// is redone exactly by the direct family.
inline uint32_t piece_capacity(double mean, double ratio)
{
    const double c = mean + 8.0 * sqrt((mean + 1.0) * ratio) + 256.0;
    uint32_t cap = (uint32_t)((uint64_t)(c + 1.0) + 1) & ~1u;
    // never a stride that is a multiple of 16 KiB: regions read side by side at a large power-of-two stride share their HBM
    // channels (a 128-KiB unit stride made seg_count 54 times slower, dk_accum_create)
    if (cap % 2048 == 0) cap += 16;
    return cap;
}

inline double segment_ratio(const dk_engine *e)
{
    const int m = e->opt.multiplicity_hint;
    return m > 0 ? std::min(64.0, std::max(2.0, (double)m + 1.0)) : 64.0;
}

// scan_part geometry: 2 = 512 threads x 16 positions, two workgroups per CU (default); 6 = 1024 x 16, one
// per CU, from 256 level-1 bins (2^16 segments) on -- with 256-512 level-1 bins the 8192-record tile leaves 16-32
// records per run and half-empty level-1 pieces, which the 16384-record tile and half as many
// workgroups repair (2^37 bits: 61 -> 76 Gk-mers/s).  Option "scan_variant" forces one (1, 3, 4, 5: experiments).
inline int scan_variant_threads(int v) { return v == 2 || v == 3 ? 512 : v == 4 ? 256 : v == 5 ? 128 : 1024; }
inline int scan_variant(const dk_engine *e, int b1, bool windowed)
{
    const int forced = windowed ? 0 : e->opt.scan_variant;       // the windowed scan is built for the two default shapes
    return forced ? forced : b1 >= 8 ? 6 : 2;
}

// KmerCounter (no set): the segments are only counting units, so their number follows the batch, not the
// filter -- about 5 K records each, which one seg_count workgroup holds in registers (46 K records per
// segment at the filter's 2^15 segments took 750 ms at configs[1], 2^18 segments take 9)
inline int count_segments_log2(const dk_engine *e, uint64_t n_records)
{
    const uint64_t per_seg = e->opt.count_seg > 0 ? (uint64_t)e->opt.count_seg : 5000ULL;
    int T = 1;
    while (T < MAX_SEG_BITS && (n_records >> T) > per_seg) T++;       // above 18 bits: three partition levels
    return T;
}

inline int set_segment_bits(const dk_engine *e) { return (int)e->cfg.filter_log2_bits - 9 - SEG_LOG2_BLOCKS; }

// Sub-segment split: with 2^19 segments to route to (a 2^38-bit set, or one of two hash windows of a 2^39-bit one) two
// multisplit levels reach 2^18 regions and the segment kernels take the last bit, which saves the third pass over
// the records (16 bytes per record of HBM traffic) for one re-read of a region by the workgroup of the sibling
// segment.  Option "sub_split": 0 = this rule, 1..3 = force, 9 = never.
inline int pick_sub_bits(const dk_engine *e, int T_local)
{
    const int o = e->opt.sub_split;
    if (o == 9) return 0;
    if (o >= 1 && o <= 3) return T_local - o >= 1 ? o : 0;
    // (one bit only: four workgroups re-reading a region cost seg_insert more than the third pass -- 50.7 vs 34.8 + 19.7 ms
    // per 48 M reads at 2^39 bits)
    const bool wide = e->cfg.k > 32;
    const int two = MAX_BIN_BITS + (wide ? MAX_BIN_BITS : (e->opt.repart_bits > 0 ? e->opt.repart_bits : MAX_BIN_BITS2));
    return T_local == two + 1 ? 1 : 0;
}

// T_override > 0: number of segment bits to use instead of the filter's.  wbits > 0: only the records of one
// hash window (1 / 2^wbits of them) are partitioned, over the T - wbits segment bits below the window's.
// sbits > 0 (insert / accumulate against a set): the partition stops sbits bits short of the 64-KiB segments, the segment
// kernels resolve them (PieceList::sbits); p->T, n_seg, cap2 then describe the regions.
inline bool make_plan(const dk_engine *e, const dk_reads *r, BucketPlan *p, int T_override = 0, int wbits = 0, int sbits = 0)
{
    const bool wide = e->cfg.k > 32;
    p->sbits = sbits;
    p->T = (T_override > 0 ? T_override : set_segment_bits(e)) - wbits - sbits;
    if (p->T < 1 || p->T > MAX_SEG_BITS) return false;
    p->b3 = 0;
    p->capA = 0;
    const int bits2 = wide ? MAX_BIN_BITS : (e->opt.repart_bits > 0 ? e->opt.repart_bits : MAX_BIN_BITS2);   // k > 32: 512-thread repart
    if (p->T > MAX_BIN_BITS + bits2 || (e->opt.force_l3 && p->T >= 3)) {
        // three levels: thirds of T; the coarse regions (b1 + b2 bits) index the grid's y dimension
        p->b1 = p->T / 3;
        p->b2 = (p->T - p->b1) / 2;
        p->b3 = p->T - p->b1 - p->b2;
    } else {
        p->b1 = (p->T + e->opt.b1_up) / 2;
        if (p->b1 > MAX_BIN_BITS) p->b1 = MAX_BIN_BITS;
        if (p->T - p->b1 > bits2) p->b1 = p->T - bits2;
        p->b2 = p->T - p->b1;
    }
    int v = scan_variant(e, p->b1, wbits > 0);
    if (!wide && !p->b3 && (1 << p->b1) > scan_variant_threads(v)) {
        // a forced geometry with fewer threads than level-1 bins (the bin scan is one thread per bin): move bits to level 2
        int t = 0;
        while ((2 << t) <= scan_variant_threads(v)) t++;
        if (p->T - t > bits2) v = scan_variant(e, p->b1, true);   // cannot: fall back to the automatic geometry
        else { p->b1 = t; p->b2 = p->T - t; }
    }
    p->variant = v;
    p->p1 = 1u << p->b1;
    p->p2 = 1u << p->b2;
    p->n_seg = 1ULL << p->T;
    const uint64_t n_all = r->n_windows && r->n_windows < r->n_bases ? r->n_windows : r->n_bases;
    // a window holds 1 / 2^wbits of the hashes (uniform), plus every copy of the heavy k-mers that fall into it
    const double n_exp = (double)n_all / (double)(1ULL << wbits);
    p->n_max = wbits ? (uint64_t)(n_exp + 8.0 * sqrt(n_exp * 64.0) + 65536.0) : n_all;
    if (p->n_max > n_all) p->n_max = n_all;
    // 16-byte records (k > 32): 512 threads x 8 positions so that the LDS stage stays at 64 KiB
    p->tile = wide ? 512 * 8 : v == 2 ? 512 * 16 : v == 3 ? 512 * 8 : v == 4 ? 256 * 16 : v == 5 ? 128 * 16 : v == 6 ? 1024 * 16 : 1024 * 8;
    const int blocks_per_cu = wide ? 2 : v == 1 ? 1 : v == 3 ? 4 : v == 4 ? 4 : v == 5 ? 6 : v == 6 ? 1 : 2;
    const uint64_t n_tiles = (r->n_bases + p->tile - 1) / p->tile;
    if (n_tiles > 0xFFFFFFFFULL) return false;
    p->G = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(n_tiles, 1), (uint64_t)e->n_cu * blocks_per_cu);
    // expected piece size from the largest share a producer can get (tiles are dealt round-robin,
    // so shares differ by at most one tile)
    const uint64_t tiles_per_wg = (n_tiles + p->G - 1) / p->G;
    const double share1 = std::min((double)p->n_max, (double)(tiles_per_wg * (uint64_t)p->tile) / (double)(1ULL << wbits));
    const double m1 = share1 / (double)p->p1;
    const double m2 = (double)p->n_max / (double)p->n_seg;
    if (m1 * 2 + 1e6 >= 4.0e9 || m2 * 2 + 1e6 >= 4.0e9) return false;    // u32 cursors
    const double ratio2 = segment_ratio(e);
    p->capw = piece_capacity(m1, 4.0);
    p->cap2 = piece_capacity(m2, ratio2);
    if (p->b3) {
        const double mA = (double)p->n_max / (double)(1ULL << (p->b1 + p->b2));
        if (mA * 2 + 1e6 >= 4.0e9) return false;
        p->capA = piece_capacity(mA, ratio2);
    }
    // piece stride = an odd multiple of 128 B past a 4-KiB boundary: the workgroups of repart read the same
    // tile of neighbouring pieces at the same time, and strides near a large power of two pile those reads
    // onto few HBM channels (measured 2 % on the whole pass)
    p->capw = (p->capw + 511) / 512 * 512 + 16;
    return true;
}

// AUTO mode: the bucketed family costs ~9-12 ps per position plus one sweep of the set (~0.24 ps per byte at
// 4.1-4.8 TB/s), the direct family ~54 ps per position (one random 64-B block per k-mer plus the global count
// table); measured crossover near 175 bytes of set per position (2^40 bits against a 12.8 M-read batch: 23.6 vs
// 14.7 Gk-mers/s).  Below a few million positions the fixed launch and sync costs of five kernels decide.
inline bool bucketed_pays(const dk_engine *e, uint64_t n_bases, int wbits = 0)
{
    const uint64_t filter_bytes = ((1ULL << e->cfg.filter_log2_bits) / 8) >> wbits;
    const int T = set_segment_bits(e) - wbits;
    if (T < 1 || T > MAX_SEG_BITS) return false;
    return filter_bytes >= (32ULL << 20) && n_bases >= (4ULL << 20) && n_bases * 128 >= filter_bytes;
}

inline unsigned repart_grid(uint64_t blocks_per_bin, uint64_t n_bins) { return (unsigned)(blocks_per_bin * n_bins); }

template <class R>
struct BucketBufs {
    R *a = nullptr, *b = nullptr;             // level-1 pieces / regions; which one ends up holding the segments' records
    R *rec = nullptr, *scratch = nullptr;     // depends on the number of levels: rec = final records, scratch = the other (absent lists)
    uint32_t *cursorA = nullptr;              // three levels: fill of the coarse regions
    uint32_t *cnt = nullptr;                  // cnt1 [p1 * G] | cursor2 [n_seg] | miss_cnt [n_seg]
    uint32_t *cnt1 = nullptr, *cursor2 = nullptr, *miss_cnt = nullptr;
    R *ovf = nullptr;                         // overflow records
    uint64_t ovf_cap = 0;
    R *ovf_miss = nullptr, *extra = nullptr;  // probe: absent overflow records, then sorted by segment
    uint32_t *extra_idx = nullptr;            // seg_hist [n_seg] | extra_off [n_seg + 1] | fill [n_seg]
    uint32_t *fine_cursor = nullptr;          // fill of the finer counting units (big batches, see bucketed_probe_t)
    R *fine = nullptr;                        // their records, when the space of the probed records is too small
};

template <class R>
inline void free_bufs(dk_engine *e, BucketBufs<R> &B)
{
    pool_free(e, B.a);
    pool_free(e, B.b);
    pool_free(e, B.cnt);
    pool_free(e, B.ovf);
    pool_free(e, B.ovf_miss);
    pool_free(e, B.extra);
    pool_free(e, B.extra_idx);
    pool_free(e, B.fine_cursor);
    pool_free(e, B.fine);
    B = BucketBufs<R>();
}

// scan_part + repart (+ repart): afterwards B.rec / B.cursor2 hold every record of the batch (of the hash window
// widx of 2^wbits, when wbits > 0) grouped by segment, except the records that did not fit, which are in B.ovf
// (Counters::n_ovf of them).  need_scratch: a second segment-sized buffer for the absent lists (per-batch probe).
template <bool WIDE>
inline dk_status bucketed_partition(dk_engine *e, const dk_reads *r, const BucketPlan &p,
                                    BucketBufs<typename RecOf<WIDE>::type> &B, int wbits = 0, uint32_t widx = 0,
                                    bool need_scratch = true)
{
    using R = typename RecOf<WIDE>::type;
    const uint64_t seg_recs = p.n_seg * (uint64_t)p.cap2;
    // 128 bytes between the pieces of consecutive level-1 bins: a workgroup of scan_part writes to 2^b1 frontiers that are
    // G * capw records apart, always a multiple of 4 KiB, so all of them sat on the same few HBM channels at any moment
    // (configs[1]: scan_part 4.75 -> 4.35 ms on one box, no difference on others)
    const uint32_t l1_skew = 128u / (uint32_t)sizeof(R);
    const uint64_t lvl1_recs = (uint64_t)p.p1 * (p.G * (uint64_t)p.capw + l1_skew);
    const uint64_t n_coarse = p.b3 ? 1ULL << (p.b1 + p.b2) : 0;
    const uint64_t coarse_recs = n_coarse * p.capA;
    // two levels: a = level-1 pieces (then the absent lists), b = segments.  three: a = level 1, then segments; b = coarse (then absent lists)
    const uint64_t a_recs = p.b3 ? std::max(seg_recs, lvl1_recs) : std::max(need_scratch ? seg_recs : 0, lvl1_recs);
    const uint64_t b_recs = p.b3 ? std::max(need_scratch ? seg_recs : 0, coarse_recs) : seg_recs;
    DK_TRY(pool_alloc(e, a_recs * sizeof(R), (void **)&B.a));
    DK_TRY(pool_alloc(e, b_recs * sizeof(R), (void **)&B.b));
    const uint64_t n1 = (uint64_t)p.p1 * p.G;
    DK_TRY(pool_alloc(e, (n1 + 2 * p.n_seg + n_coarse) * 4, (void **)&B.cnt));
    B.cnt1 = B.cnt;
    B.cursor2 = B.cnt + n1;
    B.miss_cnt = B.cursor2 + p.n_seg;
    B.cursorA = B.miss_cnt + p.n_seg;
    B.rec = p.b3 ? B.a : B.b;
    B.scratch = p.b3 ? B.b : B.a;
    if (n_coarse) DK_HIP(e, hipMemsetAsync(B.cursorA, 0, n_coarse * 4, e->stream));
    B.ovf_cap = std::max<uint64_t>(1ULL << 20, p.n_max / 8);
    DK_TRY(pool_alloc(e, B.ovf_cap * sizeof(R), (void **)&B.ovf));
    DK_HIP(e, hipMemsetAsync(B.cursor2, 0, p.n_seg * 4, e->stream));
    const OvfList<R> ovf{B.ovf, &e->d_ctr->n_ovf, B.ovf_cap};

    StreamView sv;
    sv.bases = r->d_bases;
    sv.mask = r->d_mask;
    sv.n_bases = r->n_bases;
    sv.n_bwords = (r->n_bases + 31) / 32;
    sv.n_mwords = (r->n_bases + 63) / 64;
    const uint32_t n_tiles = (uint32_t)((r->n_bases + p.tile - 1) / p.tile);
#define DK_SCAN_LAUNCH(TH, PT, W, WIN)                                                                                    \
    scan_part_kernel<TH, PT, W, WIDE, WIN><<<p.G, TH, 0, e->stream>>>(sv, (int)e->cfg.k, (int)e->cfg.canonical,           \
                                                                      e->cfg.seed, p.b1, p.capw, B.a, B.cnt1, n_tiles,    \
                                                                      ovf, e->d_ctr, wbits, widx, l1_skew)
    // level 2: the level-1 pieces -> the segments' regions, or (three levels) -> 2^(b1+b2) coarse regions
#define DK_REPART_LAUNCH(TH, PT, W)                                                                       \
    do {                                                                                                  \
        const uint32_t tpp = (p.capw + TH * PT - 1) / (TH * PT);                                           \
        repart_kernel<TH, PT, W, R><<<repart_grid(p.G * tpp, p.p1), TH, 0, e->stream>>>(                   \
            B.a, B.cnt1, p.G, p.capw, tpp, wbits + p.b1, p.b2, p.b3 ? p.capA : p.cap2, B.b,                 \
            p.b3 ? B.cursorA : B.cursor2, ovf, e->d_ctr, !e->opt.repart_plain && p.p1 % 8 == 0, l1_skew);     \
    } while (0)
    // level 3: every coarse region is one "piece" (G = 1) of the same kernel, split by b3 more bits
#define DK_REPART3_LAUNCH(TH, PT, W)                                                                      \
    do {                                                                                                  \
        const uint32_t tpp = (p.capA + TH * PT - 1) / (TH * PT);                                           \
        repart_kernel<TH, PT, W, R><<<repart_grid(tpp, 1u << (p.b1 + p.b2)), TH, 0, e->stream>>>(          \
            B.b, B.cursorA, 1u, p.capA, tpp, wbits + p.b1 + p.b2, p.b3, p.cap2, B.a, B.cursor2, ovf, e->d_ctr,  \
            !e->opt.repart_plain && ((1u << (p.b1 + p.b2)) % 8 == 0));                                        \
    } while (0)
    if constexpr (WIDE) {
        if (wbits) DK_SCAN_LAUNCH(512, 8, 4, true);
        else DK_SCAN_LAUNCH(512, 8, 4, false);
        DK_HIP(e, hipGetLastError());
        stage_mark(e, "scan_part");
        DK_REPART_LAUNCH(512, 8, 8);
        if (p.b3) {
            DK_HIP(e, hipGetLastError());
            stage_mark(e, "repart");
            DK_REPART3_LAUNCH(512, 8, 8);
        }
    } else {
        if (wbits) {
            if (p.variant == 6) DK_SCAN_LAUNCH(1024, 16, 4, true);
            else DK_SCAN_LAUNCH(512, 16, 4, true);
        } else {
            switch (p.variant) {
            case 1: DK_SCAN_LAUNCH(1024, 8, 4, false); break;
            case 2: DK_SCAN_LAUNCH(512, 16, 4, false); break;
            case 3: DK_SCAN_LAUNCH(512, 8, 8, false); break;
            case 4: DK_SCAN_LAUNCH(256, 16, 4, false); break;
            case 5: DK_SCAN_LAUNCH(128, 16, 3, false); break;
            case 6: DK_SCAN_LAUNCH(1024, 16, 4, false); break;
            default: DK_SCAN_LAUNCH(1024, 8, 8, false); break;
            }
        }
        DK_HIP(e, hipGetLastError());
        stage_mark(e, "scan_part");
        if (e->opt.repart_variant == 1) DK_REPART_LAUNCH(1024, 16, 4);
        else DK_REPART_LAUNCH(1024, 8, 8);
        if (p.b3) {
            DK_HIP(e, hipGetLastError());
            stage_mark(e, "repart");
            DK_REPART3_LAUNCH(1024, 8, 8);
        }
    }
#undef DK_SCAN_LAUNCH
#undef DK_REPART_LAUNCH
#undef DK_REPART3_LAUNCH
    DK_HIP(e, hipGetLastError());
    stage_mark(e, p.b3 ? "repart3" : "repart");
    return DK_OK;
}

// copy the device counters to the host; the absent tallies of the segment kernels (Counters::shard) are folded
// into n_absent on both sides, so every later copy sees one consistent number
inline dk_status sync_counters(dk_engine *e, const char *what)
{
    hipError_t h = hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream);
    if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
    if (h != hipSuccess) return fail(e, DK_ERR_HIP, "%s failed: %s", what, hipGetErrorString(h));
    unsigned long long sh = 0;
    for (int i = 0; i < COUNTER_SHARDS; i++) sh += e->h_ctr->shard[i];
    if (sh) {
        e->h_ctr->n_absent += sh;
        memset(e->h_ctr->shard, 0, sizeof e->h_ctr->shard);
        h = hipMemcpyAsync(&e->d_ctr->n_absent, &e->h_ctr->n_absent, 8, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess) h = hipMemsetAsync(e->d_ctr->shard, 0, sizeof e->h_ctr->shard, e->stream);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);     // the copy reads h_ctr, which the caller goes on to edit
        if (h != hipSuccess) return fail(e, DK_ERR_HIP, "%s failed: %s", what, hipGetErrorString(h));
    }
    if (e->h_ctr->n_overflow)
        return fail(e, DK_ERR_OVERFLOW, "bucket overflow (%llu records)", (unsigned long long)e->h_ctr->n_overflow);
    return DK_OK;
}

// (A persistent walk of the set kernels -- two workgroups per CU stepping through the segments with the next segment
// in flight to registers while the current one is probed -- measured no better for seg_insert (33.8-35.2 vs 34.1 ms per
// 48 M reads at 2^39 bits) and worse for seg_probe (16 vs 9.5 ms: the prefetch registers spill at the 64 VGPRs that two
// workgroups per CU allow); one workgroup per segment it is.)
// Returns DK_ERR_OVERFLOW when even the overflow list overflowed: the caller then runs the direct
// family on the whole batch, which is exact (OR is idempotent, records already inserted do no harm).
template <bool WIDE>
inline dk_status bucketed_insert_t(dk_engine *e, dk_set *s, const dk_reads *r)
{
    using R = typename RecOf<WIDE>::type;
    BucketPlan p;
    const int T_full = set_segment_bits(e);
    if (!make_plan(e, r, &p, 0, 0, pick_sub_bits(e, T_full))) return fail(e, DK_ERR_UNSUPPORTED, "no bucketed plan for this geometry");
    BucketBufs<R> B;
    dk_status st = bucketed_partition<WIDE>(e, r, p, B, 0, 0, false);
    if (st == DK_OK) {
        PieceList<R> pl{B.rec, B.cursor2, 1, p.cap2, nullptr, nullptr};
        pl.sbits = p.sbits;
        pl.sub_shift = 64 - T_full;
        const unsigned n_seg = (unsigned)(p.n_seg << p.sbits);
        if (s->exact)
            seg_exact_insert_kernel<R><<<n_seg, SEG_THREADS, 0, e->stream>>>(s->d_words, pl, T_full, e->d_ctr);
        else
            seg_insert_kernel<R><<<n_seg, SEG_THREADS, 0, e->stream>>>(
                s->d_words, pl, (int)e->cfg.n_hashes, 64 - T_full - SEG_LOG2_BLOCKS);
        hipError_t h = hipGetLastError();
        if (h == hipSuccess) {
            stage_mark(e, s->exact ? "seg_exact_insert" : "seg_insert");
            // overflow records (normally none): the kernel reads their number from device memory
            const OvfList<R> ovf{B.ovf, &e->d_ctr->n_ovf, B.ovf_cap};
            ovf_insert_kernel<R><<<e->n_cu * 2, DIRECT_BLOCK, 0, e->stream>>>(
                s->d_words, ovf, (int)e->cfg.filter_log2_bits - 9, (int)e->cfg.n_hashes, s->exact ? T_full : 0, e->d_ctr);
            h = hipGetLastError();
        }
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_insert launch failed: %s", hipGetErrorString(h));
    }
    if (st == DK_OK) st = sync_counters(e, "bucketed insert");
    if (st == DK_OK && e->h_ctr->n_ovf) stage_mark(e, "ovf_insert");
    free_bufs(e, B);
    return st;
}

// (Sub-segment split: one workgroup per REGION that keeps the region's records in registers and stages the sibling
// segments one after the other reads the records once -- 47 GB instead of 59 GB per launch at 2^39 bits, two hash
// windows -- and still measured slower than sibling workgroups, 12.0 vs 9.6 ms: with two workgroups per CU, many short
// independent workgroups overlap their load / probe phases better than fewer, longer ones.)
// the membership kernel of one batch over the n_seg segments from seg_base on (the set's kind and hash count pick the
// instance); s == nullptr is only valid with ACC: every record is absent
template <class R, bool ACC>
inline hipError_t launch_seg_probe(dk_engine *e, dk_set *s, const PieceList<R> &list, uint64_t n_seg, int T_full,
                                   uint64_t seg_base, const MissOut<R> &mo)
{
    const int blk_shift = 64 - T_full - SEG_LOG2_BLOCKS;
    if (s && s->exact)
        seg_exact_probe_kernel<R, ACC><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(s->d_words, list, T_full, seg_base, mo, e->d_ctr);
    else if (s && e->cfg.n_hashes == 4)
        seg_probe_kernel<R, 4, ACC><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(s->d_words, list, 4, blk_shift, seg_base, mo, e->d_ctr);
    else
        seg_probe_kernel<R, 0, ACC><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(
            s ? s->d_words : nullptr, list, s ? (int)e->cfg.n_hashes : 0, blk_shift, seg_base, mo, e->d_ctr);
    return hipGetLastError();
}

// Count the records of `list` unit by unit into res (seg_count): n_units units whose hashes share the top Tc bits
// (unit_base + local index), n_absent records in all, extra_room more entries per region.
// The table is sized for every record being distinct when min_count == 1.  With min_count > 1 few records
// survive (a whole-genome child keeps ~1.5 % of its absent occurrences at min_count 2): the table is then sized for
// an eighth of the upper bound n_absent / min_count, and if a region runs out the kernel has still tallied what
// each region needs (region_fill), so the count is redone once with exactly that much room.
template <bool WIDE>
inline dk_status bucketed_count_stage(dk_engine *e, const PieceList<typename RecOf<WIDE>::type> &list, uint64_t n_units,
                                      int Tc, uint64_t unit_base, uint64_t n_absent, uint64_t extra_room, uint32_t min_count,
                                      dk_result *res, uint64_t size_records = 0)
{
    if (!n_absent) return DK_OK;
    const uint64_t per_seg = n_absent / n_units;
    auto launch = [&](uint64_t region_cap) -> hipError_t {
        if (per_seg >= (e->opt.cnt_big > 0 ? (uint64_t)e->opt.cnt_big : (WIDE ? 3500u : 7000u))) {
            // big segments: 1024 threads hold 8K (k > 32) / 16K records in registers, 256-Kbit bitmaps
            const unsigned cgrid = (unsigned)std::min<uint64_t>(n_units, (uint64_t)e->n_cu * 2);
            seg_count_kernel<1024, 2048, 8192, WIDE><<<cgrid, 1024, 0, e->stream>>>(
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);
        } else if (per_seg >= (WIDE ? 1300u : (uint64_t)(e->opt.cnt_mid > 0 ? e->opt.cnt_mid : 3600))) {
            const unsigned cgrid = (unsigned)std::min<uint64_t>(n_units, (uint64_t)e->n_cu * 6);
            seg_count_kernel<512, 2048, 2048, WIDE><<<cgrid, 512, 0, e->stream>>>(
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);
        } else if (per_seg >= (WIDE ? 600u : 1200u)) {
            // 256 threads hold 2K (k > 32) / 4K records: 2^17 segments at configs[1] leave ~1.6 K absent records each
            const unsigned cgrid = (unsigned)std::min<uint64_t>(n_units, (uint64_t)e->n_cu * 12);
            seg_count_kernel<256, 1024, 1024, WIDE><<<cgrid, 256, 0, e->stream>>>(
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);
        } else {
            const unsigned cgrid = (unsigned)std::min<uint64_t>(n_units, (uint64_t)e->n_cu * 32);
            seg_count_kernel<128, 512, 256, WIDE><<<cgrid, 128, 0, e->stream>>>(
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);
        }
        return hipGetLastError();
    };
    // RESULT_REGIONS output regions, each with its own fill counter; segments are dealt to the
    // regions round-robin, so the regions fill evenly (12.5 % + 64 Ki entries of slack each);
    // overflow records may all sit in one segment, hence the extra room for them
    const uint64_t used_regions = std::min<uint64_t>(RESULT_REGIONS, n_units);
    // size_records (accumulators: their capacity): the optimistic table is sized from it instead of from n_absent, so that
    // every counting pass of one accumulator asks the pool for the same block and none of them waits for hipMalloc
    const uint64_t bound = min_count > 1 ? std::max(n_absent, size_records) / min_count / 8 : n_absent;
    uint64_t region_cap = bound / used_regions + bound / (8 * used_regions) + 65536 + extra_room;
    dk_status st = DK_OK;
    for (int attempt = 0; attempt < 2; attempt++) {
        st = pool_alloc(e, region_cap * RESULT_REGIONS * 8, (void **)&res->d_lo);
        if (st == DK_OK && WIDE) st = pool_alloc(e, region_cap * RESULT_REGIONS * 8, (void **)&res->d_hi);
        if (st == DK_OK) st = pool_alloc(e, region_cap * RESULT_REGIONS * 4, (void **)&res->d_cnt);
        if (st != DK_OK) return st;
        const hipError_t h = launch(region_cap);
        if (h != hipSuccess) return fail(e, DK_ERR_HIP, "seg_count launch failed: %s", hipGetErrorString(h));
        stage_mark(e, attempt ? "seg_count_redo" : "seg_count");
        st = sync_counters(e, "bucketed count");
        uint64_t need = 0;
        for (int j = 0; j < RESULT_REGIONS; j++) need = std::max<uint64_t>(need, e->h_ctr->region_fill[j]);
        if (st != DK_ERR_OVERFLOW || attempt || min_count == 1 || need <= region_cap) break;
        // the optimistic table was too small: same grid, same walk -- every region receives exactly what it was tallied
        pool_free(e, res->d_lo);
        pool_free(e, res->d_hi);
        pool_free(e, res->d_cnt);
        res->d_lo = res->d_hi = nullptr;
        res->d_cnt = nullptr;
        region_cap = need;
        e->h_ctr->n_distinct = e->h_ctr->n_overflow = 0;
        hipError_t h2 = hipMemsetAsync(e->d_ctr->region_fill, 0, sizeof e->h_ctr->region_fill, e->stream);
        if (h2 == hipSuccess) h2 = hipMemsetAsync(&e->d_ctr->n_distinct, 0, 8, e->stream);
        if (h2 == hipSuccess) h2 = hipMemsetAsync(&e->d_ctr->n_overflow, 0, 8, e->stream);
        if (h2 != hipSuccess) return fail(e, DK_ERR_HIP, "counter reset failed: %s", hipGetErrorString(h2));
    }
    if (st == DK_OK) {
        res->n_regions = RESULT_REGIONS;
        res->region_cap = region_cap;
        res->n = 0;
        for (int j = 0; j < RESULT_REGIONS; j++) {
            res->region_n[j] = e->h_ctr->region_fill[j];
            res->n += res->region_n[j];
        }
        e->h_ctr->n_emitted = res->n;
    }
    return st;
}

template <bool WIDE>
inline dk_status bucketed_probe_t(dk_engine *e, dk_set *s, const dk_reads *r, dk_result *res)
{
    using R = typename RecOf<WIDE>::type;
    BucketPlan p;
    const uint64_t n_max = r->n_windows && r->n_windows < r->n_bases ? r->n_windows : r->n_bases;
    if (!make_plan(e, r, &p, s ? 0 : count_segments_log2(e, n_max)))
        return fail(e, DK_ERR_UNSUPPORTED, "no bucketed plan for this geometry");
    BucketBufs<R> B;
    dk_status st = bucketed_partition<WIDE>(e, r, p, B);
    PieceList<R> list{B.rec, B.cursor2, 1, p.cap2, nullptr, nullptr};
    // (a seg_count workgroup holds 16 K records of 8 bytes, 8 K of 16; k > 32: units of ~3.3 K records for the 512-thread
    // count kernel, whose registers hold 4 K: 26.6 ms against 33.6 ms with units of 1.6 K and 45 ms with the 1024-thread
    // kernel on the configs[4] batch)
    const uint64_t split_above = WIDE ? 7000 : 14000, split_to = e->opt.cnt_split_to > 0 ? (uint64_t)e->opt.cnt_split_to : (WIDE ? 3400 : 6000);
    int Tc = p.T;
    bool sunk_fine = false;                   // the absent records went straight into finer counting units
    if (st == DK_OK && s) {
        const uint32_t miss_cap = p.cap2;
        const MissOut<R> mo{B.scratch, miss_cap, B.miss_cnt, 0, 0, OvfList<R>{nullptr, nullptr, 0}};
        // Big batches against a small filter can leave more absent records per segment than a seg_count workgroup
        // holds in registers.  Where the segments are large enough for that, the first 64 of them are probed on their own
        // (hashes spread evenly: they tell the absent rate of the batch to a few per cent); if the rate is that high, the
        // membership kernel appends every segment's absent records to 2^u finer units by the next u hash bits, as it does
        // for an accumulator, and they are counted from there -- no second pass over the absent lists (count_split below,
        // which stays as the fallback for a unit that runs full).
        int u = 0;
        const uint64_t n_sample = 64;
        if (p.cap2 > split_above && p.n_seg > 2 * n_sample && !e->opt.sink_plain) {
            hipError_t h = launch_seg_probe<R, false>(e, s, list, n_sample, p.T, 0, mo);
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_probe launch failed: %s", hipGetErrorString(h));
            if (st == DK_OK) st = sync_counters(e, "membership sample");
            if (st == DK_OK) {
                const uint64_t est = e->h_ctr->n_absent / n_sample;
                e->h_ctr->n_absent = 0;
                h = hipMemsetAsync(&e->d_ctr->n_absent, 0, 8, e->stream);
                if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "counter reset failed: %s", hipGetErrorString(h));
                if (est > split_above) {
                    u = 1;
                    while (u < MAX_SUB_BITS && (est >> u) > split_to) u++;
                    if (p.T + u > MAX_SEG_BITS) u = 0;
                }
                if (st == DK_OK && u) {
                    const uint64_t n_fine = p.n_seg << u;
                    const double per_seg = std::min(1.1 * (double)est + 64.0, (double)e->h_ctr->n_valid / (double)p.n_seg);
                    // same stride rule as the accumulator's units (dk_accum_create): a multiple of 4 KiB plus 128 bytes
                    const uint32_t per_4k = 4096u / (uint32_t)sizeof(R), odd = 128u / (uint32_t)sizeof(R);
                    const uint32_t need = piece_capacity(per_seg / (double)(1u << u), 16.0);
                    const uint32_t cap_u = (need > odd ? (need - odd + per_4k - 1) / per_4k * per_4k : 0u) + odd;
                    R *store = B.scratch;
                    st = pool_alloc(e, n_fine * 4, (void **)&B.fine_cursor);
                    if (st == DK_OK && n_fine * (uint64_t)cap_u > p.n_seg * (uint64_t)p.cap2) {
                        st = pool_alloc(e, n_fine * (uint64_t)cap_u * sizeof(R), (void **)&B.fine);
                        store = B.fine;
                    }
                    if (st == DK_OK) {
                        h = hipMemsetAsync(B.fine_cursor, 0, n_fine * 4, e->stream);
                        // a record whose unit is full only bumps n_overflow (no overflow list): the batch is then probed again
                        // the plain way
                        const MissOut<R> mf{store, cap_u, B.fine_cursor, u, 64 - p.T - u, OvfList<R>{nullptr, &e->d_ctr->dbg[0], 0}};
                        if (h == hipSuccess) h = launch_seg_probe<R, true>(e, s, list, p.n_seg, p.T, 0, mf);
                        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_probe launch failed: %s", hipGetErrorString(h));
                    }
                    if (st == DK_OK) {
                        stage_mark(e, s->exact ? "seg_exact_probe" : "seg_probe");
                        st = sync_counters(e, "bucketed probe");
                        if (st == DK_ERR_OVERFLOW) {
                            pool_free(e, B.fine_cursor);        // (the fallback below may allocate them again)
                            pool_free(e, B.fine);
                            B.fine_cursor = nullptr;
                            B.fine = nullptr;
                            e->h_ctr->n_overflow = 0;
                            e->h_ctr->n_absent = 0;
                            h = hipMemsetAsync(&e->d_ctr->n_overflow, 0, 8, e->stream);
                            if (h == hipSuccess) h = hipMemsetAsync(&e->d_ctr->n_absent, 0, 8, e->stream);
                            st = h == hipSuccess ? DK_OK : fail(e, DK_ERR_HIP, "counter reset failed: %s", hipGetErrorString(h));
                        } else if (st == DK_OK) {
                            sunk_fine = true;
                            list = PieceList<R>{store, B.fine_cursor, 1, cap_u, nullptr, nullptr};
                            Tc = p.T + u;
                        }
                    }
                }
            }
        }
        if (st == DK_OK && !sunk_fine) {
            const hipError_t h = launch_seg_probe<R, false>(e, s, list, p.n_seg, p.T, 0, mo);
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_probe launch failed: %s", hipGetErrorString(h));
            else stage_mark(e, s->exact ? "seg_exact_probe" : "seg_probe");
            list = PieceList<R>{B.scratch, B.miss_cnt, 1, miss_cap, nullptr, nullptr};
        }
    }
    if (st == DK_OK && !sunk_fine) st = sync_counters(e, "bucketed probe");
    uint64_t n_absent = 0;
    if (st == DK_OK) {
        if (!s) e->h_ctr->n_absent = e->h_ctr->n_valid - e->h_ctr->n_ovf;   // KmerCounter: every record in a segment counts
        n_absent = e->h_ctr->n_absent;
    }
    // Counting units: with few absent records per filter segment (2^18 segments and more) up to four adjacent
    // segments are counted together -- their absent lists are the "pieces" of one unit, their hashes share the
    // top T - g bits -- so that seg_count sees ~3 K records per unit instead of a few hundred
    uint32_t unit_pieces = 1;
    if (st == DK_OK && s && !sunk_fine) {
        while (unit_pieces < (uint32_t)MAX_R && Tc > 1 && (n_absent >> Tc) < 1200) {
            Tc--;
            unit_pieces *= 2;
        }
        list.n_pieces = unit_pieces;
    }
    uint64_t n_units = 1ULL << Tc;
    // overflow records (normally none): probe them one by one, sort the absent ones by counting unit (CSR)
    // and hand them to seg_count as an extra list of their unit
    if (st == DK_OK && e->h_ctr->n_ovf) {
        const uint64_t n_ovf = e->h_ctr->n_ovf;
        st = pool_alloc(e, n_ovf * sizeof(R), (void **)&B.ovf_miss);
        if (st == DK_OK) st = pool_alloc(e, n_ovf * sizeof(R), (void **)&B.extra);
        if (st == DK_OK) st = pool_alloc(e, (3 * n_units + 1) * 4, (void **)&B.extra_idx);
        hipError_t h = hipSuccess;
        if (st == DK_OK) {
            uint32_t *hist = B.extra_idx, *off = hist + n_units, *fill = off + n_units + 1;
            const OvfList<R> ovf{B.ovf, &e->d_ctr->n_ovf, B.ovf_cap};
            h = hipMemsetAsync(B.extra_idx, 0, (3 * n_units + 1) * 4, e->stream);
            if (h == hipSuccess) {
                ovf_probe_kernel<R><<<grid_for(e, n_ovf, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                    s ? s->d_words : nullptr, ovf, (int)e->cfg.filter_log2_bits - 9, (int)e->cfg.n_hashes,
                    s && s->exact ? p.T : 0, Tc, 0, B.ovf_miss, hist, e->d_ctr);
                ovf_scan_kernel<<<1, 1024, 0, e->stream>>>(hist, off, (uint32_t)n_units);
                h = hipGetLastError();
            }
            if (h == hipSuccess) {
                stage_mark(e, "ovf_probe");
                st = sync_counters(e, "overflow probe");
            } else {
                st = fail(e, DK_ERR_HIP, "overflow probe failed: %s", hipGetErrorString(h));
            }
            if (st == DK_OK && e->h_ctr->n_ovf_miss) {
                const uint64_t n_om = e->h_ctr->n_ovf_miss;
                ovf_scatter_kernel<R><<<grid_for(e, n_om, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                    B.ovf_miss, n_om, Tc, 0, off, fill, B.extra);
                h = hipGetLastError();
                if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow scatter failed: %s", hipGetErrorString(h));
                list.extra = B.extra;
                list.extra_off = off;
                n_absent += n_om;
            }
        }
    }
    if (st == DK_OK) {
        e->h_ctr->n_absent = n_absent;           // dk_probe reports it; keep the device copy in step
        hipError_t h = hipMemcpyAsync(&e->d_ctr->n_absent, &e->h_ctr->n_absent, 8, hipMemcpyHostToDevice, e->stream);
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "counter update failed: %s", hipGetErrorString(h));
    }
    // Big batches against a small filter leave more absent records per segment than a seg_count workgroup holds
    // in registers (16 K), and its multi-chunk path is slow (150 ms at 16 K per segment).  The absent lists are
    // then split once more by the next hash bits -- the level-3 use of repart, over the absent lists instead of
    // coarse regions -- into units of ~5 K records; the space of the probed records is free for the result.
    if (st == DK_OK && s && !sunk_fine && unit_pieces == 1 && !list.extra && n_absent / p.n_seg > split_above) {
        int bs = 1;
        while (bs < MAX_BIN_BITS && (n_absent >> (p.T + bs)) > split_to) bs++;
        const uint64_t n_fine = p.n_seg << bs;
        const uint32_t cap_f = piece_capacity((double)n_absent / (double)n_fine, 16.0);   // an overflowing unit only costs the fallback
        if (p.T + bs <= MAX_SEG_BITS) {
            R *fine_out = B.rec;                  // the probed records are no longer needed
            st = pool_alloc(e, n_fine * 4, (void **)&B.fine_cursor);
            if (st == DK_OK && n_fine * (uint64_t)cap_f > p.n_seg * (uint64_t)p.cap2) {
                st = pool_alloc(e, n_fine * (uint64_t)cap_f * sizeof(R), (void **)&B.fine);
                fine_out = B.fine;
            }
            hipError_t h = hipSuccess;
            if (st == DK_OK) h = hipMemsetAsync(B.fine_cursor, 0, n_fine * 4, e->stream);
            if (st == DK_OK && h == hipSuccess) {
                // no overflow list here: a record that does not fit bumps n_overflow and the split is abandoned
                const OvfList<R> none{nullptr, &e->d_ctr->dbg[0], 0};
                constexpr int TH = WIDE ? 512 : 1024;
                const uint32_t tpp = (p.cap2 + TH * 8 - 1) / (TH * 8);
                repart_kernel<TH, 8, 8, R><<<repart_grid(tpp, p.n_seg), TH, 0, e->stream>>>(
                    B.scratch, B.miss_cnt, 1u, p.cap2, tpp, p.T, bs, cap_f, fine_out, B.fine_cursor, none, e->d_ctr);
                h = hipGetLastError();
                if (h == hipSuccess) h = hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream);
                if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
            }
            if (st == DK_OK && h != hipSuccess) st = fail(e, DK_ERR_HIP, "absent-list split failed: %s", hipGetErrorString(h));
            if (st == DK_OK) {
                stage_mark(e, "count_split");
                if (e->h_ctr->n_overflow == 0) {
                    list = PieceList<R>{fine_out, B.fine_cursor, 1, cap_f, nullptr, nullptr};
                    Tc = p.T + bs;
                    n_units = n_fine;
                } else {                                   // a unit overflowed (heavy repeats): count the unsplit lists
                    e->h_ctr->n_overflow = 0;
                    h = hipMemsetAsync(&e->d_ctr->n_overflow, 0, 8, e->stream);
                    if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "counter reset failed: %s", hipGetErrorString(h));
                }
            }
        }
    }
    if (st == DK_OK)
        st = bucketed_count_stage<WIDE>(e, list, n_units, Tc, 0, n_absent, e->h_ctr->n_ovf_miss, e->cfg.min_count, res);
    free_bufs(e, B);
    return st;
}

// ---- accumulator: one batch (dk_accum_add) ------------------------------------------------------------------
template <class R>
inline MissOut<R> accum_out(dk_engine *e, const dk_accum *a)
{
    return MissOut<R>{(R *)a->store, a->unit_cap, a->fill, a->u, 64 - a->T - a->u, OvfList<R>{(R *)a->ovf, a->d_novf, a->ovf_cap}};
}
inline int accum_unit_bits(const dk_accum *a) { return a->T + a->u; }
inline uint64_t accum_unit_base(const dk_accum *a) { return (uint64_t)a->widx << (a->T - a->wbits + a->u); }

// Partition the batch's records of the accumulator's hash window, test them against the set and append the absent
// ones to the accumulator's units.  DK_ERR_OVERFLOW with nothing appended when the partition's overflow list
// overflowed (the caller redoes the batch through the direct family); any other failure leaves the accumulator unusable.
template <bool WIDE>
inline dk_status bucketed_accum_add_t(dk_engine *e, dk_accum *a, const dk_reads *r, bool *appended)
{
    using R = typename RecOf<WIDE>::type;
    *appended = false;
    BucketPlan p;
    const int sbits = a->s ? pick_sub_bits(e, a->T - a->wbits) : 0;
    if (!make_plan(e, r, &p, a->T, a->wbits, sbits)) return fail(e, DK_ERR_UNSUPPORTED, "no bucketed plan for this geometry");
    BucketBufs<R> B;
    dk_status st = bucketed_partition<WIDE>(e, r, p, B, a->wbits, a->widx, false);
    // nothing may be appended from a batch whose partition lost records: look before the membership kernel runs
    if (st == DK_OK) st = sync_counters(e, "bucketed partition");
    if (st != DK_OK) { free_bufs(e, B); return st; }
    PieceList<R> list{B.rec, B.cursor2, 1, p.cap2, nullptr, nullptr};
    list.sbits = p.sbits;
    list.sub_shift = 64 - a->T;
    const MissOut<R> mo = accum_out<R>(e, a);
    const uint64_t seg_base = (uint64_t)a->widx << (a->T - a->wbits);
    *appended = true;
    hipError_t h = launch_seg_probe<R, true>(e, a->s, list, p.n_seg << p.sbits, a->T, seg_base, mo);
    if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_probe launch failed: %s", hipGetErrorString(h));
    else stage_mark(e, a->s && a->s->exact ? "seg_exact_probe" : a->s ? "seg_probe" : "seg_append");
    // overflow records of the partition (normally none): probe one by one, append the absent ones through global cursors
    if (st == DK_OK && e->h_ctr->n_ovf) {
        const uint64_t n_ovf = e->h_ctr->n_ovf;
        st = pool_alloc(e, n_ovf * sizeof(R), (void **)&B.ovf_miss);
        if (st == DK_OK) {
            const OvfList<R> ovf{B.ovf, &e->d_ctr->n_ovf, B.ovf_cap};
            ovf_probe_kernel<R><<<grid_for(e, n_ovf, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                a->s ? a->s->d_words : nullptr, ovf, (int)e->cfg.filter_log2_bits - 9, (int)e->cfg.n_hashes,
                a->s && a->s->exact ? a->T : 0, 1, 0, B.ovf_miss, nullptr, e->d_ctr);
            h = hipGetLastError();
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow probe failed: %s", hipGetErrorString(h));
        }
        if (st == DK_OK) st = sync_counters(e, "overflow probe");
        if (st == DK_OK && e->h_ctr->n_ovf_miss) {
            const uint64_t n_om = e->h_ctr->n_ovf_miss;
            acc_append_kernel<R><<<grid_for(e, n_om, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                B.ovf_miss, n_om, accum_unit_bits(a), accum_unit_base(a), mo, e->d_ctr);
            h = hipGetLastError();
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow append failed: %s", hipGetErrorString(h));
            else stage_mark(e, "ovf_append");
        }
    }
    if (st == DK_OK) st = sync_counters(e, "bucketed accumulate");
    if (st == DK_OK) e->h_ctr->n_absent += e->h_ctr->n_ovf_miss;
    free_bufs(e, B);
    return st;
}

inline dk_status bucketed_insert(dk_engine *e, dk_set *s, const dk_reads *r)
{
    return e->cfg.k > 32 ? bucketed_insert_t<true>(e, s, r) : bucketed_insert_t<false>(e, s, r);
}

inline dk_status bucketed_probe(dk_engine *e, dk_set *s, const dk_reads *r, dk_result *res)
{
    return e->cfg.k > 32 ? bucketed_probe_t<true>(e, s, r, res) : bucketed_probe_t<false>(e, s, r, res);
}

}  // namespace dk
