/*
 * dk_oracle.h -- CPU oracle for the denovo_kmer hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * PARITY UNPINNED: the reference snapshot (/root/reference) holds no source, no tests and
 * no golden vectors (only .gitignore:1 and .github/workflows/ci.yml:1-50, see SURVEY.md
 * section 0.1).  The files BASELINE.json names for this path -- kmer.rs (k-mer extraction,
 * canonicalisation, hashing) and counter.rs (KmerCounter / KmerSet, parent-set membership,
 * child-only emission) -- are NOT IN THE MOUNT, and there is no Rust toolchain here.  This
 * oracle therefore restates the written spec of SURVEY.md section 9 (assumptions A-1..A-9,
 * frozen in DESIGN.md section 2), not reference source lines.  Every "bit-exact" claim made
 * with it means "bit-exact against this spec".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path (denovo_kmer_amd/, include/) never links or calls it.
 *
 * The algorithms here are deliberately written differently from the HIP kernels
 * (rolling 2-bit update over ASCII here; direct bit-field extraction from packed words
 * there) so that a shared bug is unlikely.
 */
#ifndef DK_ORACLE_H
#define DK_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* A k-mer of up to 64 bases: hi holds bases beyond the low 32 (zero for k <= 32). */
typedef struct { uint64_t hi, lo; } orc_kmer;

typedef struct {
    uint64_t n_reads;      /* reads seen */
    uint64_t n_windows;    /* sum over reads of max(0, L-k+1); N-containing windows included */
    uint64_t n_valid;      /* windows without a non-ACGT base */
    uint64_t n_absent;     /* probe only: valid windows whose k-mer is absent from the filter */
    uint64_t n_distinct;   /* probe only: distinct absent k-mers (before min_count) */
} orc_stats;

/* spec A-4: murmur3 64-bit finaliser and its use as the k-mer hash */
uint64_t orc_fmix64(uint64_t x);
uint64_t orc_hash_kmer(orc_kmer km, int k, uint64_t seed);

/* spec A-1/A-2/A-3/A-5: windows of one read.  out/valid hold max(0,len-k+1) entries;
 * returns that count.  valid[i]=0 when window i contains a non-ACGT byte (out[i] is then 0). */
uint64_t orc_read_kmers(const uint8_t *seq, uint64_t len, int k, int canonical,
                        orc_kmer *out, uint8_t *valid);

/* blocked-Bloom geometry: block index and the n_hashes bit positions (0..511) inside it */
void orc_bloom_positions(uint64_t h, int log2_bits, int n_hashes,
                         uint64_t *block, uint32_t *bits /* n_hashes */);

/* Insert every valid k-mer of the reads (concatenated ASCII, offsets[n_reads+1]) into
 * filter (2^log2_bits bits as little-endian u64 words).  Returns 0. */
int orc_bloom_insert_reads(uint64_t *filter, int log2_bits, int n_hashes, uint64_t seed,
                           int k, int canonical,
                           const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                           orc_stats *stats);

/* Probe the reads against filter; absent k-mers are counted.  Results sorted ascending by
 * (hi, lo); only k-mers with count >= min_count are written.  cap = capacity of the output
 * arrays (n_windows always suffices).  Returns number written, or -1 if cap is too small. */
int64_t orc_bloom_probe_reads(const uint64_t *filter, int log2_bits, int n_hashes, uint64_t seed,
                              int k, int canonical, uint32_t min_count,
                              const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                              orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                              orc_stats *stats);

/* Same result with the reads split over n_threads OpenMP threads (bench.py's cpu_baseline). */
/* wall seconds of the last orc_bloom_probe_reads_mt call: [0] probe (parallel), [1] sort (parallel), [2] merge (serial) */
void orc_last_phase_seconds(double *out);
int64_t orc_bloom_probe_reads_mt(const uint64_t *filter, int log2_bits, int n_hashes, uint64_t seed,
                                 int k, int canonical, uint32_t min_count,
                                 const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                                 orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                                 orc_stats *stats, int n_threads);

/* Exact-set semantics (what a HashSet-based KmerSet would give): child k-mers
 * absent from the union of the parent reads.  Same output conventions as the Bloom probe. */
int64_t orc_exact_child_only(int k, int canonical, uint32_t min_count,
                             const uint8_t *pseq, const uint64_t *poffsets, uint64_t p_reads,
                             const uint8_t *cseq, const uint64_t *coffsets, uint64_t c_reads,
                             orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                             orc_stats *stats);

/* Per-sample counting (KmerCounter semantics): distinct canonical k-mers with counts, sorted. */
int64_t orc_count_reads(int k, int canonical,
                        const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                        orc_kmer *out_kmers, uint32_t *out_counts, uint64_t cap,
                        orc_stats *stats);

/* Packed read-batch format of include/denovo_kmer.h restated: bases[] 32 per u64 word MSB
 * first, mask[] 64 per u64 word MSB first, one masked separator base after every read.
 * n_words_bases = ceil(total/32), n_words_mask = ceil(total/64), total = sum(L_i + 1).
 * Returns total (bases incl. separators). */
uint64_t orc_pack_reads(const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                        uint64_t *bases, uint64_t *mask);

/* packed fixed-length reads -> ASCII (n_reads * read_len bytes) */
void orc_unpack_fixed(const uint64_t *bases, const uint64_t *mask, uint64_t n_reads, uint32_t read_len,
                      uint8_t *out);

/* Synthetic trio generator (spec: DESIGN.md section 7; counter-based, stateless).
 * sample: 0 = parent 1, 1 = parent 2, 2 = child.  Writes read_len ASCII bytes. */
typedef struct {
    uint64_t seed;
    uint64_t genome_len;
    uint32_t read_len;
    double snv_rate, denovo_rate, err_rate, n_rate;
    uint64_t xover_block;
} orc_synth_cfg;
void orc_synth_read(const orc_synth_cfg *cfg, int sample, uint64_t read_idx, uint8_t *out);
uint64_t orc_synth_mix(uint64_t seed, uint64_t stream, uint64_t idx);

#ifdef __cplusplus
}
#endif
#endif
