#define _POSIX_C_SOURCE 200809L
/*
 * dk_oracle.c -- CPU oracle (plain C) for the denovo_kmer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED -- see dk_oracle.h: kmer.rs / counter.rs (named by BASELINE.json) are not in
 * /root/reference, so each function cites the spec clause (SURVEY.md section 9 = DESIGN.md
 * section 2) it restates instead of a reference file:line.
 */
#include "dk_oracle.h"
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <math.h>

typedef unsigned __int128 u128;

/* ---- spec A-4: hash = murmur3 fmix64 (Appleby, MurmurHash3.cpp, public domain) ---------- */
uint64_t orc_fmix64(uint64_t x)
{
    x ^= x >> 33;
    x *= 0xff51afd7ed558ccdULL;
    x ^= x >> 33;
    x *= 0xc4ceb9fe1a85ec53ULL;
    x ^= x >> 33;
    return x;
}

/* h = fmix64(lo ^ t); t = seed for k <= 32, t = seed ^ fmix64(hi + golden) for k > 32 */
uint64_t orc_hash_kmer(orc_kmer km, int k, uint64_t seed)
{
    uint64_t t = seed;
    if (k > 32) t ^= orc_fmix64(km.hi + 0x9E3779B97F4A7C15ULL);
    return orc_fmix64(km.lo ^ t);
}

/* ---- spec A-1: A=0 C=1 G=2 T=3, case-insensitive; everything else is "N" --------------- */
static inline int base_code(uint8_t c)
{
    switch (c) {
    case 'A': case 'a': return 0;
    case 'C': case 'c': return 1;
    case 'G': case 'g': return 2;
    case 'T': case 't': return 3;
    default: return -1;
    }
}

/* ---- spec A-1/A-2/A-3/A-5: rolling forward / reverse-complement, canonical = min -------- */
uint64_t orc_read_kmers(const uint8_t *seq, uint64_t len, int k, int canonical,
                        orc_kmer *out, uint8_t *valid)
{
    if (k < 1 || k > 64 || len < (uint64_t)k) return 0;
    const u128 mask = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    u128 fwd = 0, rc = 0;
    uint64_t run = 0;                         /* consecutive ACGT bases ending here */
    uint64_t nw = len - (uint64_t)k + 1;
    for (uint64_t i = 0; i < len; i++) {
        int c = base_code(seq[i]);
        if (c < 0) { run = 0; fwd = 0; rc = 0; }
        else {
            fwd = ((fwd << 2) | (u128)c) & mask;                  /* first base most significant */
            rc = (rc >> 2) | ((u128)(3 - c) << (2 * (k - 1)));    /* complement enters at the top */
            run++;
        }
        if (i + 1 >= (uint64_t)k) {
            uint64_t w = i + 1 - (uint64_t)k;
            if (run >= (uint64_t)k) {
                u128 km = (canonical && rc < fwd) ? rc : fwd;
                out[w].hi = (uint64_t)(km >> 64);
                out[w].lo = (uint64_t)km;
                valid[w] = 1;
            } else {
                out[w].hi = out[w].lo = 0;
                valid[w] = 0;
            }
        }
    }
    return nw;
}

/* ---- blocked Bloom: 512-bit blocks; block from the top hash bits; double hashing inside -- */
void orc_bloom_positions(uint64_t h, int log2_bits, int n_hashes, uint64_t *block, uint32_t *bits)
{
    int lb = log2_bits - 9;                       /* log2(number of blocks) */
    *block = (lb > 0) ? (h >> (64 - lb)) : 0;
    uint32_t a = (uint32_t)(h & 511);
    uint32_t d = (uint32_t)((h >> 9) & 511) | 1u;
    for (int j = 0; j < n_hashes; j++) bits[j] = (a + (uint32_t)j * d) & 511;
}

static inline void bloom_set(uint64_t *filter, uint64_t h, int log2_bits, int n_hashes)
{
    uint64_t blk; uint32_t bits[16];
    orc_bloom_positions(h, log2_bits, n_hashes, &blk, bits);
    for (int j = 0; j < n_hashes; j++)
        filter[blk * 8 + (bits[j] >> 6)] |= 1ULL << (bits[j] & 63);
}

static inline int bloom_test(const uint64_t *filter, uint64_t h, int log2_bits, int n_hashes)
{
    uint64_t blk; uint32_t bits[16];
    orc_bloom_positions(h, log2_bits, n_hashes, &blk, bits);
    for (int j = 0; j < n_hashes; j++)
        if (!((filter[blk * 8 + (bits[j] >> 6)] >> (bits[j] & 63)) & 1)) return 0;
    return 1;
}

static uint64_t max_read_len(const uint64_t *offsets, uint64_t n_reads)
{
    uint64_t m = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        uint64_t l = offsets[r + 1] - offsets[r];
        if (l > m) m = l;
    }
    return m;
}

/* spec A-7/A-8: every valid k-mer of every parent read goes in; no count threshold */
int orc_bloom_insert_reads(uint64_t *filter, int log2_bits, int n_hashes, uint64_t seed,
                           int k, int canonical,
                           const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                           orc_stats *stats)
{
    uint64_t ml = max_read_len(offsets, n_reads);
    orc_kmer *km = (orc_kmer *)malloc((ml + 1) * sizeof(orc_kmer));
    uint8_t *va = (uint8_t *)malloc(ml + 1);
    orc_stats st = {0};
    for (uint64_t r = 0; r < n_reads; r++) {
        uint64_t l = offsets[r + 1] - offsets[r];
        uint64_t nw = orc_read_kmers(seq + offsets[r], l, k, canonical, km, va);
        st.n_reads++;
        st.n_windows += nw;
        for (uint64_t w = 0; w < nw; w++) {
            if (!va[w]) continue;
            st.n_valid++;
            bloom_set(filter, orc_hash_kmer(km[w], k, seed), log2_bits, n_hashes);
        }
    }
    free(km); free(va);
    if (stats) *stats = st;
    return 0;
}

static int cmp_kmer(const void *a, const void *b)
{
    const orc_kmer *x = (const orc_kmer *)a, *y = (const orc_kmer *)b;
    if (x->hi != y->hi) return x->hi < y->hi ? -1 : 1;
    if (x->lo != y->lo) return x->lo < y->lo ? -1 : 1;
    return 0;
}

/* sort + run-length encode a list of k-mers; apply min_count; returns number written or -1 */
static int64_t rle_emit(orc_kmer *list, uint64_t n, uint32_t min_count,
                        orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                        uint64_t *n_distinct)
{
    qsort(list, n, sizeof(orc_kmer), cmp_kmer);
    uint64_t nout = 0, nd = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t j = i + 1;
        while (j < n && cmp_kmer(&list[i], &list[j]) == 0) j++;
        uint64_t c = j - i;
        nd++;
        if (c >= min_count) {
            if (nout >= cap) return -1;
            out_kmers[nout] = list[i];
            out_counts[nout] = c > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)c;
            nout++;
        }
        i = j;
    }
    if (n_distinct) *n_distinct = nd;
    return (int64_t)nout;
}

/* spec A-6/A-9: child windows probed against the parent filter; absent ones counted.
 * n_threads > 1 splits the reads into contiguous ranges over OpenMP threads (each with its own
 * absent list); the final sort + run-length encode is serial.  Results do not depend on n_threads. */
typedef struct { orc_kmer *list; uint64_t n, cap; orc_stats st; } probe_part;

static void probe_range(const uint64_t *filter, int log2_bits, int n_hashes, uint64_t seed, int k, int canonical,
                        const uint8_t *seq, const uint64_t *offsets, uint64_t r0, uint64_t r1, uint64_t ml,
                        probe_part *out)
{
    orc_kmer *km = (orc_kmer *)malloc((ml + 1) * sizeof(orc_kmer));
    uint8_t *va = (uint8_t *)malloc(ml + 1);
    out->cap = 1024;
    out->n = 0;
    out->list = (orc_kmer *)malloc(out->cap * sizeof(orc_kmer));
    memset(&out->st, 0, sizeof out->st);
    for (uint64_t r = r0; r < r1; r++) {
        uint64_t l = offsets[r + 1] - offsets[r];
        uint64_t nw = orc_read_kmers(seq + offsets[r], l, k, canonical, km, va);
        out->st.n_reads++;
        out->st.n_windows += nw;
        for (uint64_t w = 0; w < nw; w++) {
            if (!va[w]) continue;
            out->st.n_valid++;
            if (bloom_test(filter, orc_hash_kmer(km[w], k, seed), log2_bits, n_hashes)) continue;
            out->st.n_absent++;
            if (out->n == out->cap) { out->cap *= 2; out->list = (orc_kmer *)realloc(out->list, out->cap * sizeof(orc_kmer)); }
            out->list[out->n++] = km[w];
        }
    }
    free(km); free(va);
}

/* wall seconds of the phases of the last orc_bloom_probe_reads_mt call: [0] extraction + hashing + filter probe
 * (parallel over reads), [1] per-thread sort of the absent lists (parallel), [2] serial multi-way merge + counting.
 * bench.py reports them apart: only [0] is the membership path proper. */
static double g_phase_s[3];
void orc_last_phase_seconds(double *out) { out[0] = g_phase_s[0]; out[1] = g_phase_s[1]; out[2] = g_phase_s[2]; }

static double wall_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int64_t orc_bloom_probe_reads_mt(const uint64_t *filter, int log2_bits, int n_hashes, uint64_t seed,
                                 int k, int canonical, uint32_t min_count,
                                 const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                                 orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                                 orc_stats *stats, int n_threads)
{
    if (n_threads < 1) n_threads = 1;
    uint64_t ml = max_read_len(offsets, n_reads);
    probe_part *parts = (probe_part *)calloc((size_t)n_threads, sizeof(probe_part));
    const double t_start = wall_s();
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (int t = 0; t < n_threads; t++) {
        uint64_t r0 = n_reads * (uint64_t)t / (uint64_t)n_threads, r1 = n_reads * (uint64_t)(t + 1) / (uint64_t)n_threads;
        probe_range(filter, log2_bits, n_hashes, seed, k, canonical, seq, offsets, r0, r1, ml, &parts[t]);
    }
    const double t_probe = wall_s();
    /* every thread sorts its own absent list; one linear multi-way merge then counts runs */
#pragma omp parallel for num_threads(n_threads) schedule(static, 1)
    for (int t = 0; t < n_threads; t++) qsort(parts[t].list, parts[t].n, sizeof(orc_kmer), cmp_kmer);
    const double t_sort = wall_s();
    orc_stats st = {0};
    for (int t = 0; t < n_threads; t++) {
        st.n_reads += parts[t].st.n_reads;
        st.n_windows += parts[t].st.n_windows;
        st.n_valid += parts[t].st.n_valid;
        st.n_absent += parts[t].st.n_absent;
    }
    uint64_t *head = (uint64_t *)calloc((size_t)n_threads, sizeof(uint64_t));
    int64_t nout = 0;
    int have_cur = 0;
    orc_kmer cur = {0, 0};
    uint64_t cur_n = 0;
    for (;;) {
        int best = -1;
        for (int t = 0; t < n_threads; t++) {
            if (head[t] >= parts[t].n) continue;
            if (best < 0 || cmp_kmer(&parts[t].list[head[t]], &parts[best].list[head[best]]) < 0) best = t;
        }
        if (best >= 0 && have_cur && cmp_kmer(&parts[best].list[head[best]], &cur) == 0) {
            cur_n++;
            head[best]++;
            continue;
        }
        if (have_cur) {                       /* close the finished run */
            st.n_distinct++;
            if (cur_n >= min_count) {
                if ((uint64_t)nout >= cap) { nout = -1; break; }
                out_kmers[nout] = cur;
                out_counts[nout] = cur_n > 0xFFFFFFFFULL ? 0xFFFFFFFFu : (uint32_t)cur_n;
                nout++;
            }
        }
        if (best < 0) break;
        cur = parts[best].list[head[best]++];
        cur_n = 1;
        have_cur = 1;
    }
    g_phase_s[0] = t_probe - t_start;
    g_phase_s[1] = t_sort - t_probe;
    g_phase_s[2] = wall_s() - t_sort;
    for (int t = 0; t < n_threads; t++) free(parts[t].list);
    free(parts);
    free(head);
    if (stats) *stats = st;
    return nout;
}

int64_t orc_bloom_probe_reads(const uint64_t *filter, int log2_bits, int n_hashes, uint64_t seed,
                              int k, int canonical, uint32_t min_count,
                              const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                              orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                              orc_stats *stats)
{
    return orc_bloom_probe_reads_mt(filter, log2_bits, n_hashes, seed, k, canonical, min_count, seq, offsets,
                                    n_reads, out_kmers, out_counts, cap, stats, 1);
}

/* collect all valid k-mers of a read set into a malloc'd list */
static orc_kmer *collect_kmers(int k, int canonical, const uint8_t *seq, const uint64_t *offsets,
                               uint64_t n_reads, uint64_t *n_out, orc_stats *st)
{
    uint64_t ml = max_read_len(offsets, n_reads);
    orc_kmer *km = (orc_kmer *)malloc((ml + 1) * sizeof(orc_kmer));
    uint8_t *va = (uint8_t *)malloc(ml + 1);
    uint64_t lcap = 1024, ln = 0;
    orc_kmer *list = (orc_kmer *)malloc(lcap * sizeof(orc_kmer));
    for (uint64_t r = 0; r < n_reads; r++) {
        uint64_t l = offsets[r + 1] - offsets[r];
        uint64_t nw = orc_read_kmers(seq + offsets[r], l, k, canonical, km, va);
        st->n_reads++;
        st->n_windows += nw;
        for (uint64_t w = 0; w < nw; w++) {
            if (!va[w]) continue;
            st->n_valid++;
            if (ln == lcap) { lcap *= 2; list = (orc_kmer *)realloc(list, lcap * sizeof(orc_kmer)); }
            list[ln++] = km[w];
        }
    }
    free(km); free(va);
    *n_out = ln;
    return list;
}

/* spec A-6 companion: exact set membership (sorted parent list + binary search) */
int64_t orc_exact_child_only(int k, int canonical, uint32_t min_count,
                             const uint8_t *pseq, const uint64_t *poffsets, uint64_t p_reads,
                             const uint8_t *cseq, const uint64_t *coffsets, uint64_t c_reads,
                             orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                             orc_stats *stats)
{
    orc_stats pst = {0}, st = {0};
    uint64_t np = 0, nc = 0;
    orc_kmer *pl = collect_kmers(k, canonical, pseq, poffsets, p_reads, &np, &pst);
    qsort(pl, np, sizeof(orc_kmer), cmp_kmer);
    orc_kmer *cl = collect_kmers(k, canonical, cseq, coffsets, c_reads, &nc, &st);
    uint64_t na = 0;
    for (uint64_t i = 0; i < nc; i++) {
        if (np && bsearch(&cl[i], pl, np, sizeof(orc_kmer), cmp_kmer)) continue;
        cl[na++] = cl[i];
    }
    st.n_absent = na;
    int64_t nout = rle_emit(cl, na, min_count, out_kmers, out_counts, cap, &st.n_distinct);
    free(pl); free(cl);
    if (stats) *stats = st;
    return nout;
}

/* KmerCounter semantics: distinct k-mers of one sample with counts */
int64_t orc_count_reads(int k, int canonical,
                        const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                        orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                        orc_stats *stats)
{
    orc_stats st = {0};
    uint64_t n = 0;
    orc_kmer *l = collect_kmers(k, canonical, seq, offsets, n_reads, &n, &st);
    st.n_absent = n;
    int64_t nout = rle_emit(l, n, 1, out_kmers, out_counts, cap, &st.n_distinct);
    free(l);
    if (stats) *stats = st;
    return nout;
}

/* ---- packed read-batch format (include/denovo_kmer.h "dk read batch") restated ----------- */
uint64_t orc_pack_reads(const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                        uint64_t *bases, uint64_t *mask)
{
    uint64_t p = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        for (uint64_t i = offsets[r]; i <= offsets[r + 1]; i++) {
            int sep = (i == offsets[r + 1]);
            int c = sep ? -1 : base_code(seq[i]);
            if ((p & 31) == 0) bases[p >> 5] = 0;
            if ((p & 63) == 0) mask[p >> 6] = 0;
            if (c >= 0) bases[p >> 5] |= (uint64_t)c << (62 - 2 * (p & 31));
            else mask[p >> 6] |= 1ULL << (63 - (p & 63));
            p++;
        }
    }
    return p;
}

/* inverse of the packer for fixed-length reads (bench.py: the GPU generates the sample, the oracle
 * reads it back as ASCII): out[r * read_len + j] for read r, position j; separators are skipped */
void orc_unpack_fixed(const uint64_t *bases, const uint64_t *mask, uint64_t n_reads, uint32_t read_len,
                      uint8_t *out)
{
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < (int64_t)n_reads; r++) {
        uint64_t p = (uint64_t)r * (read_len + 1);
        uint8_t *o = out + (uint64_t)r * read_len;
        for (uint32_t j = 0; j < read_len; j++, p++) {
            const int flag = (int)((mask[p >> 6] >> (63 - (p & 63))) & 1);
            const int code = (int)((bases[p >> 5] >> (62 - 2 * (p & 31))) & 3);
            o[j] = flag ? 'N' : (uint8_t)ACGT[code];
        }
    }
}

/* ---- synthetic trio generator -------------------------------------- */
static inline uint64_t splitmix(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    uint64_t z = x;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t orc_synth_mix(uint64_t seed, uint64_t stream, uint64_t idx)
{
    return splitmix(splitmix(seed ^ (stream * 0xD1342543DE82EF95ULL)) + idx);
}

static inline uint64_t rate_thr(double rate)
{
    if (rate <= 0) return 0;
    if (rate >= 1) return ~0ULL;
    return (uint64_t)ldexp(rate, 64);
}

static inline int alt_of(uint64_t r) { return 1 + (int)(((r & 0xFFFF) * 3) >> 16); }

enum { S_GENOME = 1, S_SNV = 2, S_XOVER = 6, S_DENOVO = 8, S_READ = 16, S_ERR = 32, S_NN = 48 };

static int hap_base(const orc_synth_cfg *c, int hid, uint64_t pos)
{
    int g = (int)(orc_synth_mix(c->seed, S_GENOME, pos) & 3);
    uint64_t r = orc_synth_mix(c->seed, S_SNV + hid, pos);
    if (r < rate_thr(c->snv_rate)) g = (g + alt_of(r)) & 3;
    return g;
}

static int sample_base(const orc_synth_cfg *c, int sample, int which, uint64_t pos)
{
    if (sample < 2) return hap_base(c, 2 * sample + which, pos);
    /* child: haplotype `which` is inherited from parent `which`, switching between that
     * parent's two haplotypes at crossover-block boundaries */
    uint64_t blk = pos / c->xover_block;
    int sel = (int)(orc_synth_mix(c->seed, S_XOVER + which, blk) & 1);
    int b = hap_base(c, 2 * which + sel, pos);
    if (which == 0) {
        uint64_t r = orc_synth_mix(c->seed, S_DENOVO, pos);
        if (r < rate_thr(c->denovo_rate)) b = (b + alt_of(r)) & 3;
    }
    return b;
}

void orc_synth_read(const orc_synth_cfg *c, int sample, uint64_t read_idx, uint8_t *out)
{
    static const char ACGT[4] = {'A', 'C', 'G', 'T'};
    uint64_t L = c->read_len;
    uint64_t u = orc_synth_mix(c->seed, S_READ + sample, read_idx);
    int which = (int)(u & 1), strand = (int)((u >> 1) & 1);
    uint64_t span = c->genome_len - L + 1;
    uint64_t start = (uint64_t)(((u128)splitmix(u) * span) >> 64);
    for (uint64_t j = 0; j < L; j++) {
        uint64_t pos = strand ? start + L - 1 - j : start + j;
        int b = sample_base(c, sample, which, pos);
        if (strand) b = 3 - b;
        uint64_t e = orc_synth_mix(c->seed, S_ERR + sample, read_idx * L + j);
        if (e < rate_thr(c->err_rate)) b = (b + alt_of(e)) & 3;
        uint64_t n = orc_synth_mix(c->seed, S_NN + sample, read_idx * L + j);
        out[j] = (n < rate_thr(c->n_rate)) ? 'N' : (uint8_t)ACGT[b];
    }
}
