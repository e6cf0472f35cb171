// dk_kernels_direct.h -- "direct" kernel family: one thread per stream position, the filter is
// touched in HBM in place.  Used for small batches (where sweeping the whole filter would cost
// more than the random accesses), for k > 32, for the overflow records of the bucketed family,
// and as the A/B baseline of the bucketed family.
//
// Roofline: HBM-bound on random 64-B filter blocks; algorithmic bytes per
// k-mer = 3L/(8(L-k+1)) [2-bit bases + 1-bit mask] + 64 [one filter block] (+64 write-back on insert).
#pragma once
#include "dk_device.h"

namespace dk {

constexpr int DIRECT_BLOCK = 256;
constexpr uint32_t SLOT_EMPTY = 0xFFFFFFFFu;
constexpr int RESULT_REGIONS = 32;      // output regions of the bucketed count kernel (one fill counter each)
constexpr int COUNTER_SHARDS = 32;      // per-workgroup tallies are spread over this many words (one word takes ~10^8 atomics/s)

struct StreamView {
    const uint64_t *bases;
    const uint64_t *mask;
    uint64_t n_bases;      // positions incl. separators
    uint64_t n_bwords;     // ceil(n_bases / 32)
    uint64_t n_mwords;     // ceil(n_bases / 64)
};

struct FilterView {
    unsigned long long *words;   // 2^log2_bits / 64 words
    int log2_blocks;             // log2_bits - 9
    int n_hashes;
    uint64_t seed;
    int exact_T;                 // 0: blocked Bloom filter; > 0: exact table with 2^exact_T segments (dk_device.h)
};

struct Counters {               // device-side statistics of one operation
    unsigned long long n_valid;
    unsigned long long n_absent;
    unsigned long long n_cand;      // candidates appended (direct probe)
    unsigned long long n_distinct;
    unsigned long long n_emitted;
    unsigned long long n_overflow;  // bucketed: records diverted to the overflow list
    unsigned long long n_ovf;       // bucketed: records appended to the overflow list
    unsigned long long n_ovf_miss;  // bucketed: overflow records absent from the filter
    unsigned long long n_set_full;  // exact set: keys that found no free slot in their segment
    unsigned long long n_in_window; // windowed scan (dk_accum_add): valid windows whose hash lies in the window
    unsigned long long dbg[8];      // DK_STAMPS diagnostic builds only: per-phase cycle sums
    unsigned long long region_fill[32];   // bucketed seg_count: entries written to each output region
    unsigned long long shard[32];         // bucketed seg_probe: absent records, tallied per workgroup (folded into n_absent on the host)
    unsigned long long n_sink_drop;       // accumulate: absent records that found neither room in their unit nor in the accumulator's overflow list
    unsigned long long fail_mark;         // slab-wise partition: 2^32 - 1 - (first slab whose partition dropped records); 0 = none did
};

// ---- ASCII -> packed stream ------------------------------------------------------------------
// one thread per 64 output positions (two bases words, one mask word)
__global__ void __launch_bounds__(DIRECT_BLOCK)
pack_ascii_kernel(const uint8_t *__restrict__ seq, const uint64_t *__restrict__ offsets,
                  uint64_t n_reads, uint64_t n_bases, uint64_t *__restrict__ bases,
                  uint64_t *__restrict__ mask)
{
    const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = chunk * 64;
    if (p0 >= n_bases) return;
    // read holding p0: largest r with offsets[r] + r <= p0
    uint64_t lo = 0, hi = n_reads;   // invariant: start(lo) <= p0, answer in [lo, hi)
    while (hi - lo > 1) {
        const uint64_t mid = lo + (hi - lo) / 2;
        if (offsets[mid] + mid <= p0) lo = mid; else hi = mid;
    }
    uint64_t r = lo;
    uint64_t rs = offsets[r], re = offsets[r + 1];
    uint64_t j = p0 - (rs + r);              // position inside read r (== L_r means separator)
    uint64_t b0 = 0, b1 = 0, m = 0;
    for (int i = 0; i < 64; i++) {
        const uint64_t p = p0 + i;
        if (p >= n_bases) break;
        bool flag;
        uint32_t code = 0;
        if (j == re - rs) {                  // separator after read r
            flag = true;
            r++;
            if (r < n_reads) { rs = re; re = offsets[r + 1]; }
            j = 0;
        } else {
            const uint32_t c = seq[rs + j] & 0xDFu;
            flag = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
            code = ((c >> 1) ^ (c >> 2)) & 3u;
            j++;
        }
        if (flag) m |= 1ULL << (63 - i);
        else if (i < 32) b0 |= (uint64_t)code << (62 - 2 * i);
        else b1 |= (uint64_t)code << (62 - 2 * (i - 32));
    }
    mask[chunk] = m;
    bases[2 * chunk] = b0;
    if (p0 + 32 < n_bases) bases[2 * chunk + 1] = b1;
}

// flag <- 0 unless every position that is stride - 1 modulo stride is flagged in the mask (separator, or N): what the
// window-major scan relies on (dk_bucket_scan.h); one thread per read
__global__ void __launch_bounds__(DIRECT_BLOCK)
verify_stride_kernel(const uint64_t *__restrict__ mask, uint64_t n_reads, uint32_t stride, uint32_t *flag)
{
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t p = r * stride + (stride - 1);
    if (!((mask[p >> 6] >> (63 - (p & 63))) & 1ULL)) *flag = 0;
}

// ---- synthetic trio reads, generated straight into the packed format ----
struct SynthParams {
    uint64_t seed, genome_len, span;   // span = genome_len - read_len + 1
    uint32_t read_len, xover_log2;
    uint64_t snv_thr, denovo_thr, err_thr, n_thr;
    int sample;
    uint64_t first_read, n_reads;
};

__host__ __device__ __forceinline__ uint64_t splitmix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t stream_key(uint64_t seed, uint64_t stream)
{
    return splitmix(seed ^ (stream * 0xD1342543DE82EF95ULL));
}
__device__ __forceinline__ int alt_of(uint64_t r) { return 1 + (int)(((r & 0xFFFF) * 3) >> 16); }

struct SynthKeys { uint64_t genome, snv[4], xover[2], denovo, read, err, nn; };

__device__ __forceinline__ int synth_hap_base(const SynthParams &c, const SynthKeys &k, int hid, uint64_t pos)
{
    int g = (int)(splitmix(k.genome + pos) & 3);
    const uint64_t r = splitmix(k.snv[hid] + pos);
    if (r < c.snv_thr) g = (g + alt_of(r)) & 3;
    return g;
}

__global__ void __launch_bounds__(DIRECT_BLOCK)
synth_kernel(SynthParams c, uint64_t n_bases, uint64_t *__restrict__ bases, uint64_t *__restrict__ mask)
{
    const uint64_t chunk = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t p0 = chunk * 64;
    if (p0 >= n_bases) return;
    SynthKeys k;
    k.genome = stream_key(c.seed, 1);
    for (int h = 0; h < 4; h++) k.snv[h] = stream_key(c.seed, 2 + h);
    k.xover[0] = stream_key(c.seed, 6);
    k.xover[1] = stream_key(c.seed, 7);
    k.denovo = stream_key(c.seed, 8);
    k.read = stream_key(c.seed, 16 + c.sample);
    k.err = stream_key(c.seed, 32 + c.sample);
    k.nn = stream_key(c.seed, 48 + c.sample);

    const uint64_t L = c.read_len, stride = L + 1;
    uint64_t r = p0 / stride, j = p0 % stride;
    uint64_t start = 0;
    int which = 0, strand = 0;
    bool have = false;
    uint64_t b0 = 0, b1 = 0, m = 0;
    for (int i = 0; i < 64; i++) {
        if (p0 + i >= n_bases) break;
        bool flag = false;
        int b = 0;
        if (j == L) {
            flag = true;
        } else {
            const uint64_t ridx = c.first_read + r;
            if (!have) {
                const uint64_t u = splitmix(k.read + ridx);
                which = (int)(u & 1);
                strand = (int)((u >> 1) & 1);
                start = __umul64hi(splitmix(u), c.span);
                have = true;
            }
            const uint64_t pos = strand ? start + L - 1 - j : start + j;
            if (c.sample < 2) {
                b = synth_hap_base(c, k, 2 * c.sample + which, pos);
            } else {
                const uint64_t blk = pos >> c.xover_log2;
                const int sel = (int)(splitmix(k.xover[which] + blk) & 1);
                b = synth_hap_base(c, k, 2 * which + sel, pos);
                if (which == 0) {
                    const uint64_t d = splitmix(k.denovo + pos);
                    if (d < c.denovo_thr) b = (b + alt_of(d)) & 3;
                }
            }
            if (strand) b = 3 - b;
            const uint64_t e = splitmix(k.err + ridx * L + j);
            if (e < c.err_thr) b = (b + alt_of(e)) & 3;
            const uint64_t n = splitmix(k.nn + ridx * L + j);
            if (n < c.n_thr) flag = true;
        }
        if (flag) m |= 1ULL << (63 - i);
        else if (i < 32) b0 |= (uint64_t)b << (62 - 2 * i);
        else b1 |= (uint64_t)b << (62 - 2 * (i - 32));
        if (++j == stride) { j = 0; r++; have = false; }
    }
    mask[chunk] = m;
    bases[2 * chunk] = b0;
    if (p0 + 32 < n_bases) bases[2 * chunk + 1] = b1;
}

// ---- block-buffered append -----------------------------------------------------------------------
// Lanes push records into an LDS buffer; when it is nearly full the block reserves a range of the
// global list with ONE atomic and copies the buffer out coalesced.  A single global counter bumped
// once per wave saturates near 10^8 atomics/s (that alone cost 100+ ms at 1.5 G windows).
template <bool WIDE, bool AUX>
struct BlockAppend {
    static constexpr int CAP = 2048;
    uint64_t lo[CAP];
    uint64_t hi[WIDE ? CAP : 1];
    uint32_t aux[AUX ? CAP : 1];
    uint32_t n;
    unsigned long long gbase;

    __device__ __forceinline__ void init()
    {
        if (threadIdx.x == 0) n = 0;
        __syncthreads();
    }
    // every lane of the wave calls push (pred selects the lanes that append)
    __device__ __forceinline__ void push(bool pred, uint64_t vlo, uint64_t vhi, uint32_t vaux)
    {
        const uint64_t b = __ballot(pred);
        if (!b) return;
        const int leader = __ffsll((long long)b) - 1;
        uint32_t base = 0;
        if (lane_id() == leader) base = atomicAdd(&n, (uint32_t)__popcll(b));
        base = __shfl(base, leader);
        if (pred) {
            const uint32_t i = base + (uint32_t)popc_below(b);
            lo[i] = vlo;
            if (WIDE) hi[i] = vhi;
            if (AUX) aux[i] = vaux;
        }
    }
    // every thread of the block calls flush_if_needed once per loop iteration (block-uniform)
    __device__ __forceinline__ void flush_if_needed(bool force, unsigned long long *counter, uint64_t cap,
                                                    uint64_t *__restrict__ out_lo, uint64_t *__restrict__ out_hi,
                                                    uint32_t *__restrict__ out_aux)
    {
        __syncthreads();
        const uint32_t cnt = n;
        if (cnt == 0 || (!force && cnt + blockDim.x <= (uint32_t)CAP)) return;     // block-uniform
        if (threadIdx.x == 0) gbase = atomicAdd(counter, (unsigned long long)cnt);
        __syncthreads();
        const unsigned long long g = gbase;
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x) {
            if (g + i < cap) {
                out_lo[g + i] = lo[i];
                if (WIDE) out_hi[g + i] = hi[i];
                if (AUX) out_aux[g + i] = aux[i];
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) n = 0;
        __syncthreads();
    }
};

// ---- direct insert ----------------------------------------------------------------------------
struct GlobalWords {
    const uint64_t *w;
    uint64_t last;
    __device__ __forceinline__ uint64_t operator()(uint64_t i) const { return w[i < last ? i : last]; }
};

template <bool WIDE>
__global__ void __launch_bounds__(DIRECT_BLOCK)
insert_direct_kernel(StreamView s, FilterView f, int k, int canonical, Counters *ctr)
{
    const GlobalWords W{s.bases, s.n_bwords - 1}, M{s.mask, s.n_mwords - 1};
    uint64_t n_valid = 0, n_full = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < s.n_bases; p += stride) {
        Kmer km;
        if (!extract_kmer<WIDE>(p, k, canonical, W, M, km)) continue;
        n_valid++;
        const uint64_t h = hash_kmer<WIDE>(km, f.seed);
        if (f.exact_T) {
            if (exact_insert_global<WIDE>(f.words, f.exact_T, h, km.hi) == 2) n_full++;
            continue;
        }
        unsigned long long *blk = f.words + bloom_block(h, f.log2_blocks) * 8;
        const uint32_t a = (uint32_t)(h & 511), d = (uint32_t)((h >> 9) & 511) | 1u;
        for (int j = 0; j < f.n_hashes; j++) {
            const uint32_t bit = (a + (uint32_t)j * d) & 511;
            const unsigned long long m = 1ULL << (bit & 63);
            // test before set: with 30x coverage most k-mers are already in; a stale read only
            // costs a redundant atomic
            if (!(blk[bit >> 6] & m)) atomicOr(&blk[bit >> 6], m);
        }
    }
    n_valid = wave_sum(n_valid);
    n_full = wave_sum(n_full);
    if (lane_id() == 0 && n_valid) atomicAdd(&ctr->n_valid, (unsigned long long)n_valid);
    if (lane_id() == 0 && n_full) atomicAdd(&ctr->n_set_full, (unsigned long long)n_full);
}

template <bool WIDE>
__device__ __forceinline__ bool filter_test(const FilterView &f, uint64_t h, uint64_t hi)
{
    if (f.exact_T) return exact_contains<WIDE>(f.words, f.exact_T, h, hi);
    const unsigned long long *blk = f.words + bloom_block(h, f.log2_blocks) * 8;
    const uint32_t a = (uint32_t)(h & 511), d = (uint32_t)((h >> 9) & 511) | 1u;
    bool all = true;
    for (int j = 0; j < f.n_hashes; j++) {
        const uint32_t bit = (a + (uint32_t)j * d) & 511;
        all = all && ((blk[bit >> 6] >> (bit & 63)) & 1ULL);
    }
    return all;
}

// ---- direct probe: absent k-mers are appended to the candidate list ----------------------------
// f.words == nullptr: every valid k-mer is a candidate (KmerCounter semantics)
template <bool WIDE>
__global__ void __launch_bounds__(DIRECT_BLOCK)
probe_direct_kernel(StreamView s, FilterView f, int k, int canonical, Counters *ctr,
                    uint64_t *__restrict__ cand_lo, uint64_t *__restrict__ cand_hi, uint64_t cand_cap)
{
    __shared__ BlockAppend<WIDE, false> app;
    app.init();
    const GlobalWords W{s.bases, s.n_bwords - 1}, M{s.mask, s.n_mwords - 1};
    uint64_t n_valid = 0, n_absent = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (s.n_bases + blockDim.x - 1) / blockDim.x * blockDim.x;   // whole blocks stay in the loop
    for (uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n_round; p += stride) {
        Kmer km{0, 0};
        bool valid = false;
        if (p < s.n_bases) valid = extract_kmer<WIDE>(p, k, canonical, W, M, km);
        bool absent = false;
        if (valid) {
            n_valid++;
            absent = f.words ? !filter_test<WIDE>(f, hash_kmer<WIDE>(km, f.seed), km.hi) : true;
        }
        if (absent) n_absent++;
        app.push(absent, km.lo, km.hi, 0);
        app.flush_if_needed(false, &ctr->n_cand, cand_cap, cand_lo, cand_hi, nullptr);
    }
    app.flush_if_needed(true, &ctr->n_cand, cand_cap, cand_lo, cand_hi, nullptr);
    n_valid = wave_sum(n_valid);
    n_absent = wave_sum(n_absent);
    if (lane_id() == 0) {
        if (n_valid) atomicAdd(&ctr->n_valid, (unsigned long long)n_valid);
        if (n_absent) atomicAdd(&ctr->n_absent, (unsigned long long)n_absent);
    }
}

// ---- exact counting of the candidates: open-addressing table of candidate indices --------------
// A slot holds the index of the first candidate that claimed it; key comparison reads the
// (immutable) candidate arrays, so 64- and 128-bit keys share one lock-free protocol.
template <bool WIDE>
__global__ void __launch_bounds__(DIRECT_BLOCK)
count_insert_kernel(const uint64_t *__restrict__ cand_lo, const uint64_t *__restrict__ cand_hi,
                    uint64_t n_cand, uint32_t *slots, uint32_t *counts, int log2_cap, uint64_t seed)
{
    const uint64_t cap_mask = (1ULL << log2_cap) - 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cand; i += stride) {
        Kmer km{WIDE ? cand_hi[i] : 0, cand_lo[i]};
        uint64_t s = hash_kmer<WIDE>(km, seed) >> (64 - log2_cap);
        for (;;) {
            uint32_t cur = slots[s];
            if (cur == SLOT_EMPTY) cur = atomicCAS(&slots[s], SLOT_EMPTY, (uint32_t)i);
            if (cur == SLOT_EMPTY) break;                       // claimed
            if (cand_lo[cur] == km.lo && (!WIDE || cand_hi[cur] == km.hi)) break;
            s = (s + 1) & cap_mask;
        }
        atomicAdd(&counts[s], 1u);
    }
}

// merge of (k-mer, count) tables: same table as count_insert, the entry's count is added.  The table is built
// for one hash range at a time (pass q of 2^pbits), which bounds its size, and slots hold 64-bit candidate indices
// once there are 2^32 - 1 candidates or more.  Pass and slot come from a REMIX of the k-mer's hash, not from its top
// bits: the tables of one accumulator window all share their top hash bits, and taken as they are they would fall into
// one pass (a table sized for a 1 / 2^pbits share then overflows) and into one corner of the table.  The probe is
// bounded by the table size: a table that is full anyway reports it (n_overflow) instead of spinning, and the host
// redoes the merge with larger pass tables.
template <class IdxT> struct SlotOf { static constexpr IdxT EMPTY = (IdxT)~(IdxT)0; };

__host__ __device__ __forceinline__ uint64_t merge_mix(uint64_t h) { return (h ^ (h >> 32)) * 0x9E3779B97F4A7C15ULL; }

template <bool WIDE, class IdxT>
__global__ void __launch_bounds__(DIRECT_BLOCK)
merge_insert_kernel(const uint64_t *__restrict__ cand_lo, const uint64_t *__restrict__ cand_hi,
                    const uint32_t *__restrict__ cand_cnt, uint64_t n_cand, IdxT *slots, uint32_t *counts,
                    int log2_cap, uint64_t seed, int pbits, uint32_t q, Counters *ctr)
{
    constexpr IdxT EMPTY = SlotOf<IdxT>::EMPTY;
    const uint64_t cap_mask = (1ULL << log2_cap) - 1;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t n_lost = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_cand; i += stride) {
        Kmer km{WIDE ? cand_hi[i] : 0, cand_lo[i]};
        const uint64_t h = merge_mix(hash_kmer<WIDE>(km, seed));
        if (pbits && (uint32_t)(h >> (64 - pbits)) != q) continue;
        uint64_t s = ((h << pbits) >> (64 - log2_cap)) & cap_mask;
        bool placed = false;
        for (uint64_t tries = 0; tries <= cap_mask; tries++) {
            IdxT cur = slots[s];
            if (cur == EMPTY) {
                if constexpr (sizeof(IdxT) == 8) cur = (IdxT)atomicCAS((unsigned long long *)&slots[s], (unsigned long long)EMPTY, (unsigned long long)i);
                else cur = (IdxT)atomicCAS((unsigned int *)&slots[s], (unsigned int)EMPTY, (unsigned int)i);
            }
            if (cur == EMPTY || (cand_lo[cur] == km.lo && (!WIDE || cand_hi[cur] == km.hi))) { placed = true; break; }   // claimed, or its key
            s = (s + 1) & cap_mask;
        }
        if (!placed) { n_lost++; continue; }                    // table full: the host redoes the merge with more room
        // saturating add: a count never wraps
        const uint32_t add = cand_cnt[i];
        uint32_t old = counts[s];
        for (;;) {
            const uint32_t want = old > 0xFFFFFFFFu - add ? 0xFFFFFFFFu : old + add;
            const uint32_t seen = atomicCAS(&counts[s], old, want);
            if (seen == old) break;
            old = seen;
        }
    }
    if (n_lost) atomicAdd(&ctr->n_overflow, (unsigned long long)n_lost);
}

template <bool WIDE, class IdxT = uint32_t>
__global__ void __launch_bounds__(DIRECT_BLOCK)
count_emit_kernel(const uint64_t *__restrict__ cand_lo, const uint64_t *__restrict__ cand_hi,
                  const IdxT *__restrict__ slots, const uint32_t *__restrict__ counts,
                  uint64_t cap, uint32_t min_count, Counters *ctr,
                  uint64_t *__restrict__ out_lo, uint64_t *__restrict__ out_hi,
                  uint32_t *__restrict__ out_cnt)
{
    constexpr IdxT EMPTY = SlotOf<IdxT>::EMPTY;
    __shared__ BlockAppend<WIDE, true> app;
    app.init();
    uint64_t n_distinct = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t n_round = (cap + blockDim.x - 1) / blockDim.x * blockDim.x;
    for (uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; s < n_round; s += stride) {
        IdxT idx = EMPTY;
        uint32_t c = 0;
        if (s < cap) idx = slots[s];
        if (idx != EMPTY) { c = counts[s]; n_distinct++; }
        const bool emit = idx != EMPTY && c >= min_count;
        uint64_t vlo = 0, vhi = 0;
        if (emit) { vlo = cand_lo[idx]; if (WIDE) vhi = cand_hi[idx]; }
        app.push(emit, vlo, vhi, c);
        app.flush_if_needed(false, &ctr->n_emitted, ~0ULL, out_lo, out_hi, out_cnt);
    }
    app.flush_if_needed(true, &ctr->n_emitted, ~0ULL, out_lo, out_hi, out_cnt);
    n_distinct = wave_sum(n_distinct);
    if (lane_id() == 0 && n_distinct) atomicAdd(&ctr->n_distinct, (unsigned long long)n_distinct);
}

// ---- KmerSet::contains for explicit k-mers -----------------------------------------------------
template <bool WIDE>
__global__ void __launch_bounds__(DIRECT_BLOCK)
contains_kernel(FilterView f, const uint64_t *__restrict__ lo, const uint64_t *__restrict__ hi,
                uint64_t n, uint8_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Kmer km{WIDE ? hi[i] : 0, lo[i]};
    out[i] = filter_test<WIDE>(f, hash_kmer<WIDE>(km, f.seed), km.hi) ? 1 : 0;
}

// ---- filter utilities ---------------------------------------------------------------------------
// dst[i] |= OR_j src[j * slice_vec + i], 16 bytes per lane
__global__ void __launch_bounds__(DIRECT_BLOCK)
or_slices_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, uint64_t n_slices, uint64_t slice_vec)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < slice_vec; i += stride) {
        uint4 acc = dst[i];
        for (uint64_t j = 0; j < n_slices; j++) {
            const uint4 v = src[j * slice_vec + i];
            acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w;
        }
        dst[i] = acc;
    }
}

__global__ void __launch_bounds__(DIRECT_BLOCK)
popcount_kernel(const uint64_t *__restrict__ words, uint64_t n_words, unsigned long long *out)
{
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += stride)
        acc += (uint64_t)__popcll(words[i]);
    acc = wave_sum(acc);
    if (lane_id() == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}

// ---- exact set utilities --------------------------------------------------------------------------
// every slot of every segment <- EMPTY of its segment (k > 32: high word 0); one thread per 16 bytes
template <bool WIDE>
__global__ void __launch_bounds__(DIRECT_BLOCK)
exact_clear_kernel(unsigned long long *table, uint64_t n_words, int T)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words / 2; i += stride) {
        const uint64_t e = exact_empty((2 * i) / EXACT_SEG_WORDS, T);
        *(ulonglong2 *)(table + 2 * i) = ulonglong2{e, WIDE ? 0ULL : e};
    }
}

// number of keys in the table
template <bool WIDE>
__global__ void __launch_bounds__(DIRECT_BLOCK)
exact_count_kernel(const unsigned long long *__restrict__ table, uint64_t n_words, int T, unsigned long long *out)
{
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words / 2; i += stride) {
        const uint64_t e = exact_empty((2 * i) / EXACT_SEG_WORDS, T);
        const ulonglong2 v = *(const ulonglong2 *)(table + 2 * i);
        acc += (v.x != e) + (WIDE ? 0 : (v.y != e));
    }
    acc = wave_sum(acc);
    if (lane_id() == 0 && acc) atomicAdd(out, (unsigned long long)acc);
}

}  // namespace dk
