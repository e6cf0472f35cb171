// dk_device.h -- device-side building blocks shared by every kernel of the engine (gfx950).
//
// Spec clauses implemented here are those of DESIGN.md section 2 (= SURVEY.md section 9, A-1..A-5);
// the reference files they stand for (kmer.rs: extraction, canonicalisation, hashing) are not in
// /root/reference, so there is no file:line to cite.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dk {

constexpr int WAVE = 64;

// ---- spec A-4: murmur3 fmix64 and its inverse (the hash is a bijection on u64, which lets a
// bucket record carry the hash only and the k-mer be recovered for the rare absent ones) -----
__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

__host__ __device__ __forceinline__ uint64_t unfmix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0x9cb4b2f8129337dbULL;   // inverse of 0xc4ceb9fe1a85ec53 mod 2^64
    x ^= x >> 33;
    x *= 0x4f74430c22a54005ULL;   // inverse of 0xff51afd7ed558ccd mod 2^64
    x ^= x >> 33;
    return x;
}

struct Kmer {
    uint64_t hi, lo;
};

// t-term of the hash: seed for k <= 32; seed ^ fmix64(hi + golden) beyond
template <bool WIDE>
__host__ __device__ __forceinline__ uint64_t hash_tweak(uint64_t hi, uint64_t seed)
{
    if (WIDE) return seed ^ fmix64(hi + 0x9E3779B97F4A7C15ULL);
    return seed;
}

template <bool WIDE>
__host__ __device__ __forceinline__ uint64_t hash_kmer(Kmer km, uint64_t seed)
{
    return fmix64(km.lo ^ hash_tweak<WIDE>(km.hi, seed));
}

// reverse the order of the 32 two-bit groups of a word
__device__ __forceinline__ uint64_t rev_pairs64(uint64_t x)
{
    uint64_t r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
}

// ---- window extraction from the packed stream (spec A-1, A-3, A-5) --------------------------
// W(i) returns word i of the bases stream, M(i) word i of the mask stream; both may be asked for
// one (WIDE: two) word(s) past the one holding position p, the accessor clamps.
// Returns true when [p, p+k) holds a k-mer; km = canonical (or forward) k-mer.
template <bool WIDE, class BasesAt, class MaskAt>
__device__ __forceinline__ bool extract_kmer(uint64_t p, int k, bool canonical,
                                             BasesAt W, MaskAt M, Kmer &km)
{
    // mask: k flags from bit (p & 63) of word p >> 6, MSB first
    const uint64_t mw = p >> 6;
    const int ms = (int)(p & 63);
    uint64_t mv = M(mw) << ms;
    if (ms) mv |= M(mw + 1) >> (64 - ms);
    const bool bad = (mv >> (64 - k)) != 0;

    const uint64_t bw = p >> 5;
    const int o = 2 * (int)(p & 31);
    if (!WIDE) {
        uint64_t v = W(bw) << o;
        if (o) v |= W(bw + 1) >> (64 - o);
        const int s = 64 - 2 * k;
        const uint64_t fwd = v >> s;
        uint64_t out = fwd;
        if (canonical) {
            const uint64_t rc = (~rev_pairs64(fwd)) >> s;
            out = rc < fwd ? rc : fwd;
        }
        km.hi = 0;
        km.lo = out;
    } else {
        const uint64_t w0 = W(bw), w1 = W(bw + 1), w2 = W(bw + 2);
        uint64_t h = w0 << o, l = w1 << o;
        if (o) { h |= w1 >> (64 - o); l |= w2 >> (64 - o); }
        const int s = 128 - 2 * k;             // 0..62 for k in 33..64
        uint64_t fh = h, fl = l;
        if (s) { fl = (l >> s) | (h << (64 - s)); fh = h >> s; }
        uint64_t oh = fh, ol = fl;
        if (canonical) {
            // reverse the 64 groups of the right-aligned value, complement, re-align
            uint64_t rh = ~rev_pairs64(fl), rl = ~rev_pairs64(fh);
            if (s) { rl = (rl >> s) | (rh << (64 - s)); rh = rh >> s; }
            if (rh < fh || (rh == fh && rl < fl)) { oh = rh; ol = rl; }
        }
        km.hi = oh;
        km.lo = ol;
    }
    return !bad;
}

// ---- blocked Bloom geometry -----------------------------------------
// block = top log2_blocks bits of h; bit_j = (a + j*d) & 511, a = h & 511, d = ((h>>9)&511)|1
__host__ __device__ __forceinline__ uint64_t bloom_block(uint64_t h, int log2_blocks)
{
    return log2_blocks > 0 ? (h >> (64 - log2_blocks)) : 0;
}

// ---- exact k-mer set -----------------------------
// The same 2^n-bit array as the Bloom filter, read as open-addressing tables.  Segment s (64 KiB,
// the unit the bucketed kernels stage in LDS) holds the keys whose hash starts with the
// T = n - 19 bits of s: 8192 slots of one u64 (the hash, a bijection of the k-mer for k <= 32) or
// 4096 slots of (hash, high word) for k > 32, probed linearly inside the segment.  EMPTY and LOCKED are values whose top T bits differ from the segment's, so no key of
// the segment equals them.  A k > 32 slot is claimed EMPTY -> LOCKED, the high word is written, then
// the hash is published, so a reader that sees the hash also sees its high word.
constexpr int EXACT_SEG_WORDS = 8192;                     // u64 words per segment
template <bool WIDE> struct ExactGeom { static constexpr uint32_t SLOTS = WIDE ? 4096 : 8192; };

__host__ __device__ __forceinline__ uint64_t exact_empty(uint64_t seg, int T) { return (seg ^ 1ULL) << (64 - T); }
// Probing runs over 32-byte buckets (4 slots; 2 for k > 32) from bucket (h >> 20): one pair of
// 16-byte reads covers a whole bucket, so a lookup is one or two round trips even when its wave-mates
// sit in long clusters, and a 32-byte HBM sector is used whole.
constexpr uint32_t EXACT_BUCKETS = EXACT_SEG_WORDS / 4;
__host__ __device__ __forceinline__ uint32_t exact_bucket0(uint64_t h) { return (uint32_t)(h >> 20) & (EXACT_BUCKETS - 1); }

// read-only lookup in one segment (LDS or HBM; no insert may be running)
template <bool WIDE>
__device__ __forceinline__ bool exact_find(const unsigned long long *seg, uint64_t EMPTY, uint64_t h, uint64_t hi)
{
    uint32_t b = exact_bucket0(h);
    for (uint32_t t = 0; t < EXACT_BUCKETS; t++) {
        const ulonglong2 p = *(const ulonglong2 *)(seg + 4 * b), q = *(const ulonglong2 *)(seg + 4 * b + 2);
        // keys are never removed, so inside a bucket the key cannot sit behind an EMPTY slot
        if constexpr (WIDE) {
            if ((p.x == h && p.y == hi) || (q.x == h && q.y == hi)) return true;
            if (p.x == EMPTY || q.x == EMPTY) return false;
        } else {
            if (p.x == h || p.y == h || q.x == h || q.y == h) return true;
            if (p.x == EMPTY || p.y == EMPTY || q.x == EMPTY || q.y == EMPTY) return false;
        }
        b = (b + 1) & (EXACT_BUCKETS - 1);
    }
    return false;
}

template <bool WIDE>
__device__ __forceinline__ bool exact_contains(const unsigned long long *table, int T, uint64_t h, uint64_t hi)
{
    const uint64_t seg = h >> (64 - T);
    return exact_find<WIDE>(table + seg * EXACT_SEG_WORDS, exact_empty(seg, T), h, hi);
}

// insert-if-absent into one segment, concurrent with other inserts (never with exact_find).
// SCOPE: __HIP_MEMORY_SCOPE_WORKGROUP for a segment staged in LDS, _AGENT for the table in HBM.
// Returns 0 = already present, 1 = inserted, 2 = the segment has no free slot.
// First a plain bucket scan: most inserts are repeats (30x coverage) and end there; it stops at the
// first slot that is not a settled foreign key.  A stale value is harmless -- a key, once seen, stays,
// and a slot seen free is re-examined atomically.  From that slot on, one slot at a time with atomics;
// one loop, no inner spin: a lane that finds a slot LOCKED retries it on the next trip, by which time
// the owner -- possibly a lane of the same wave, executing the other branch of this trip -- has
// published it.
template <bool WIDE, int SCOPE>
__device__ __forceinline__ int exact_insert(unsigned long long *seg, uint64_t EMPTY, uint64_t h, uint64_t hi)
{
    constexpr uint32_t SLOTS = ExactGeom<WIDE>::SLOTS, MASK = SLOTS - 1, PER_BUCKET = WIDE ? 2 : 4;
    const unsigned long long LOCKED = EMPTY | 1ULL;
    uint32_t b = exact_bucket0(h), slot = 0;
    bool open = false;
    for (uint32_t t = 0; t < EXACT_BUCKETS && !open; t++) {
        asm volatile("" ::: "memory");                     // re-read: other lanes are inserting
        const ulonglong2 p = *(const ulonglong2 *)(seg + 4 * b), q = *(const ulonglong2 *)(seg + 4 * b + 2);
        if constexpr (WIDE) {
            if ((p.x == h && p.y == hi) || (q.x == h && q.y == hi)) return 0;
            const bool f0 = (p.x | 1ULL) == LOCKED || p.x == h, f1 = (q.x | 1ULL) == LOCKED || q.x == h;
            if (f0 || f1) { slot = PER_BUCKET * b + (f0 ? 0u : 1u); open = true; }
        } else {
            if (p.x == h || p.y == h || q.x == h || q.y == h) return 0;
            const bool f0 = p.x == EMPTY, f1 = p.y == EMPTY, f2 = q.x == EMPTY, f3 = q.y == EMPTY;
            if (f0 || f1 || f2 || f3) { slot = PER_BUCKET * b + (f0 ? 0u : f1 ? 1u : f2 ? 2u : 3u); open = true; }
        }
        b = (b + 1) & (EXACT_BUCKETS - 1);
    }
    if (!open) return 2;
    uint32_t t = 0;
    int res = -1;
    while (res < 0) {
        unsigned long long *ph = seg + (WIDE ? 2 * slot : slot);
        unsigned long long cur = __hip_atomic_load(ph, __ATOMIC_ACQUIRE, SCOPE);
        if (cur == EMPTY) {
            unsigned long long expect = EMPTY;
            __hip_atomic_compare_exchange_strong(ph, &expect, WIDE ? LOCKED : (unsigned long long)h, __ATOMIC_ACQUIRE,
                                                 __ATOMIC_ACQUIRE, SCOPE);
            cur = expect;                                  // the value found
            if (cur == EMPTY) {
                if constexpr (WIDE) {
                    __hip_atomic_store(ph + 1, (unsigned long long)hi, __ATOMIC_RELAXED, SCOPE);
                    __hip_atomic_store(ph, (unsigned long long)h, __ATOMIC_RELEASE, SCOPE);
                }
                res = 1;
                continue;
            }
        }
        if (WIDE && cur == LOCKED) continue;               // being published: look again
        if (cur == h && (!WIDE || __hip_atomic_load(ph + 1, __ATOMIC_RELAXED, SCOPE) == hi)) {
            res = 0;
        } else {
            slot = (slot + 1) & MASK;
            if (++t >= SLOTS) res = 2;
        }
    }
    return res;
}

template <bool WIDE>
__device__ __forceinline__ int exact_insert_global(unsigned long long *table, int T, uint64_t h, uint64_t hi)
{
    const uint64_t seg = h >> (64 - T);
    return exact_insert<WIDE, __HIP_MEMORY_SCOPE_AGENT>(table + seg * EXACT_SEG_WORDS, exact_empty(seg, T), h, hi);
}

// ---- wave-level helpers ---------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ int popc_below(uint64_t ballot)
{
    // number of set bits of `ballot` in lanes below this one
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(ballot >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)ballot, 0));
}

// wave-aggregated append: lanes with pred reserve consecutive slots from *counter.
// Returns this lane's slot (valid only where pred).
__device__ __forceinline__ uint64_t wave_append(bool pred, unsigned long long *counter)
{
    const uint64_t b = __ballot(pred);
    const int n = __popcll(b);
    uint64_t base = 0;
    if (n) {
        const int leader = __ffsll((long long)b) - 1;
        if (lane_id() == leader) base = atomicAdd(counter, (unsigned long long)n);
        base = __shfl(base, leader);
    }
    return base + (uint64_t)popc_below(b);
}

// Inclusive prefix sum across the 64 lanes of a wave with DPP row shifts / broadcasts: pure VALU,
// no LDS crossbar traffic (__shfl_* lowers to ds_bpermute, whose latency balloons when the LDS is
// busy, which is exactly when the multisplit kernels scan their bin counts).
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v)
{
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 -> rows 1,3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 -> rows 2,3
    return v;
}

// sum over the wave, returned to every lane (scalar broadcast of lane 63 of the scan)
__device__ __forceinline__ uint32_t wave_total(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_incl_scan(v), 63);
}

__device__ __forceinline__ uint64_t wave_sum(uint64_t v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;   // lane 0 holds the sum
}

}  // namespace dk
