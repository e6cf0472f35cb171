/*
 * denovo_kmer.h -- C ABI of the MI355X-native de-novo k-mer engine (libdenovo_kmer.so).
 *
 * Drop-in boundary for the hot path BASELINE.json names: k-mer extraction / canonicalisation /
 * hashing (reference: kmer.rs -- NOT IN MOUNT) and parent-set membership + child-only counting
 * (reference: counter.rs, types KmerCounter / KmerSet -- NOT IN MOUNT).  /root/reference holds
 * only .gitignore:1 and .github/workflows/ci.yml:1-50, so no reference FFI exists to cite; the
 * reference interface each entry point stands in for is named by type (KmerSet::insert, ...),
 * as reconstructed in SURVEY.md section 8b.  A Rust host binds this header 1:1 with an
 * `extern "C"` block; no C++ types, exceptions or callbacks cross it.
 *
 * Conventions
 *  - every call returns a dk_status (0 = DK_OK); dk_last_error(engine) gives the message
 *  - one engine = one GPU + one HIP stream; calls on one engine are serialised by the caller
 *    (Rust: Send, not Sync); engines on different devices are independent
 *  - input pointers are borrowed for the duration of the call; handles are owned by the library
 *  - k-mers: A=0 C=1 G=2 T=3, first base most significant; lo = low 64 bits, hi = bits above
 *    (hi = 0 for k <= 32); canonical = min(forward, reverse complement)
 *
 * Packed read batch ("dk_reads"), also accepted from the caller:
 *    bases : u64 words, 32 bases per word, base i of the stream at bits [62-2*(i%32), +2) of
 *            word i/32 (first base in the most significant bits)
 *    mask  : u64 words, 64 flags per word, flag i at bit 63-(i%64) of word i/64; 1 = this
 *            position is not A/C/G/T (N) or is a read separator
 *    every read is followed by exactly one separator position (mask = 1, base bits ignored);
 *    n_bases counts separators: n_bases = sum(L_i) + n_reads.
 *  A window [p, p+k) is a k-mer iff none of its k mask flags is set.
 */
#ifndef DENOVO_KMER_H
#define DENOVO_KMER_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DK_ABI_VERSION 3

typedef int32_t dk_status;
enum {
    DK_OK = 0,
    DK_ERR_INVALID_ARG = 1,
    DK_ERR_NO_DEVICE = 2,      /* no HIP device / HIP runtime failure at init: never a CPU fallback */
    DK_ERR_HIP = 3,
    DK_ERR_OOM = 4,
    DK_ERR_UNSUPPORTED = 5,
    DK_ERR_OVERFLOW = 6,
    DK_ERR_SET_FULL = 7        /* exact set: a segment has no free slot left; raise filter_log2_bits */
};

enum { DK_MODE_AUTO = 0, DK_MODE_DIRECT = 1, DK_MODE_BUCKETED = 2 };

/* What a dk_set holds in its 2^filter_log2_bits bits (both are HBM-resident and use the same kernels
 * around them):
 *  DK_SET_BLOOM  blocked Bloom filter (default): no false negatives, false positives at the
 *                configured size, any number of k-mers.
 *  DK_SET_EXACT  exact set -- what a HashSet<u64> gives: 64-KiB segments, each an open-addressing
 *                table of the k-mers whose hash starts with the segment's bits (8 bytes per slot for
 *                k <= 32, 16 bytes for k > 32).  Capacity 2^(n-6) k-mers (2^(n-7) for k > 32); size
 *                it for a load of at most ~75 %.  dk_set_insert returns DK_ERR_SET_FULL when a k-mer
 *                finds no free slot in its segment (the set then holds a subset of what was given).
 *                n_hashes is ignored. */
enum { DK_SET_BLOOM = 0, DK_SET_EXACT = 1 };

typedef struct dk_engine dk_engine;
typedef struct dk_reads dk_reads;     /* device-resident packed read batch */
typedef struct dk_set dk_set;         /* KmerSet: parent blocked-Bloom filter resident in HBM */
typedef struct dk_result dk_result;   /* k-mer -> count table (child-only set, or KmerCounter output) */
typedef struct dk_accum dk_accum;     /* child-only k-mer occurrences of many batches, counted once at the end */

typedef struct dk_config {
    uint64_t struct_size;        /* = sizeof(dk_config); versions the struct */
    uint32_t k;                  /* 1..64 */
    uint32_t canonical;          /* 1 = strand-neutral (default), 0 = forward only */
    uint32_t filter_log2_bits;   /* 20..40: filter holds 2^n bits */
    uint32_t n_hashes;           /* 1..16 bits set per k-mer inside its 512-bit block */
    uint64_t seed;
    uint32_t min_count;          /* emit child-only k-mers seen >= min_count times (>=1) */
    int32_t  device_id;
    uint32_t rank, world_size;   /* informational; sharding is done by the host */
    uint32_t mode;               /* DK_MODE_* : kernel family for insert / probe */
    uint32_t set_kind;           /* DK_SET_* : what the engine's sets hold (0 = Bloom filter) */
    void    *stream;             /* optional hipStream_t to run on; NULL = engine-owned stream */
} dk_config;

typedef struct dk_stats {
    uint64_t n_reads;
    uint64_t n_bases;       /* stream positions incl. separators */
    uint64_t n_windows;     /* sum max(0, L-k+1): the k-mers/sec denominator */
    uint64_t n_valid;       /* windows without N */
    uint64_t n_absent;      /* probe/count: valid windows absent from the set */
    uint64_t n_distinct;    /* probe/count: distinct absent k-mers (before min_count) */
    uint64_t n_emitted;     /* probe/count: k-mers in the result (after min_count) */
} dk_stats;

#define DK_MAX_STAGES 12
typedef struct dk_timings {
    uint32_t n_stages;
    float    total_ms;                  /* first event to last event of the last operation */
    float    stage_ms[DK_MAX_STAGES];   /* HIP-event time of each stage on the engine stream */
    char     stage_name[DK_MAX_STAGES][24];
} dk_timings;

/* synthetic trio generator; thresholds are rate * 2^64 */
typedef struct dk_synth_config {
    uint64_t struct_size;
    uint64_t seed;
    uint64_t genome_len;
    uint32_t read_len;
    uint32_t xover_log2;        /* crossover block = 2^xover_log2 bases */
    uint64_t snv_thr, denovo_thr, err_thr, n_thr;
} dk_synth_config;

/* ---- library ------------------------------------------------------------------------------ */
int32_t     dk_abi_version(void);
const char *dk_status_string(dk_status s);

/* ---- engine (replaces: construction of KmerCounter/KmerSet with k, counter.rs) -------------- */
dk_status   dk_engine_create(const dk_config *cfg, dk_engine **out);
void        dk_engine_destroy(dk_engine *e);
const char *dk_last_error(const dk_engine *e);          /* e may be NULL: last create error */
dk_status   dk_engine_synchronize(dk_engine *e);
dk_status   dk_engine_timings(const dk_engine *e, dk_timings *out);   /* of the last insert/probe/count */
dk_status   dk_engine_config(const dk_engine *e, dk_config *out);
/* Run-time options, validated (unknown name or value out of range: DK_ERR_INVALID_ARG); 0 restores the default.
 *   "multiplicity_hint"  expected copies of one k-mer inside ONE batch (~ the batch's coverage of the genome).
 *                        Capacity planning of the bucket regions only -- results are exact for any value, a low one
 *                        makes the overflow path likelier.  Default: up to 64 copies (a whole 30-60x sample per batch);
 *                        set ~2 for one of the ~40 batches of a whole-genome sample: the workspace shrinks ~2x.
 *   test hooks, which force a kernel geometry on inputs too small to select it:
 *   "scan_variant" 1..6, "repart_variant" 0..1, "force_l3" 0..1, "b1_up" -4..4, "count_seg" >= 64, "cnt_mid" >= 1,
 *   "sub_split" 0..3 | 9 (0 = automatic, 9 = never), "repart_plain" 0..1, "merge_pass_bits" 0..8, "merge_idx64" 0..1,
 *   "sink_plain" 0..1 (1: dk_probe never routes absent records into finer counting units while probing),
 *   "accum_unit_cap" records per counting unit of the next dk_accum_create, used when it is at least what the capacity needs,
 *   "scan_bits" 0 | 1..10 (most hash bits level 1 takes; 9 = round 2's layout), "slabs" 0..1024 (slab-wise level 2 of
 *   insert / accumulate: number of slabs, 0 = automatic from "slab_mb", the room of one slab's regions in MiB),
 *   "ovf_cap" capacity of the partition's overflow list in records, "accum_plain" 0..1 (1: 8-byte accumulator records
 *   even where 6-byte packed ones apply), "accum_min_u" 0..10 (at least 2^n counting units per segment),
 *   "mode" 0..2 (kernel family override: 0 = dk_config.mode, 1 = direct, 2 = bucketed), "l2_packed" 0..1 (6-byte records in
 *   the level-2 regions too: measured slower, kept for A/B runs), "merge_undersize" 0..10 (dk_result_merge starts with pass
 *   tables 2^n times too small: its redo path), "repart_pieces" 0..2 (level 2 reads a bin piece by piece / as the concatenation
 *   of its pieces; 0 = automatic), "scan_positions" 0..1 (1: the scan never deals windows instead of positions to its threads
 *   for batches of one read length), "kmers_plain" 0..1 (1: ordinary instead of non-temporal stores in dk_reads_kmers),
 *   "comm_staging_kb" (size of the staging buffer dk_comm_init takes, KiB; 0 = 1 GiB; at least 64 KiB per peer),
 *   "l1_layout" 0..1 (level-1 pieces bin-major / workgroup-major: measured equal, kept for A/B runs), "l1_skew" (bytes
 *   between the level-1 pieces of consecutive bins, a multiple of 16; 0 = 128) */
dk_status   dk_engine_set_option(dk_engine *e, const char *name, int64_t value);
/* What the engine did / holds, by name: "plan_levels", "plan_b1", "plan_b2", "plan_b3", "plan_sbits", "plan_slabs",
 * "plan_scan_variant", "plan_segment_bits" (the partition plan of the last bucketed operation), "pool_bytes_in_use",
 * "pool_bytes_cached", "pool_bytes_reserved", "pool_bytes_peak" (device memory).  Unknown name: DK_ERR_INVALID_ARG. */
dk_status   dk_engine_get_info(const dk_engine *e, const char *name, int64_t *value);
/* The engine keeps the device memory of finished operations in a pool for the next one (no allocation inside timed
 * work); it hands cached blocks back by itself when an allocation fails.  dk_engine_trim frees every cached block now --
 * between the parent build and the child pass of a whole-genome run, whose workspaces differ -- and reports the bytes freed. */
dk_status   dk_engine_trim(dk_engine *e, uint64_t *bytes_freed);
/* Allocate `bytes` of device memory now, as one arena that every later workspace, read batch, set and accumulator of the
 * engine is carved from (best fit, free neighbours coalesce): the one slow hipMalloc of a whole-genome run (seconds for
 * 150 GB) then happens where the host chooses -- while it opens its input files -- instead of inside the first batch.
 * May be called more than once (one more arena each).  bytes = 0: hand back every arena nothing lives in. */
dk_status   dk_engine_reserve(dk_engine *e, uint64_t bytes);

/* ---- read batches (replaces: the &[u8] read sequences handed to kmer.rs by the BAM loop) --- */
/* ASCII reads, concatenated, offsets[n_reads+1]; packed on the GPU */
dk_status dk_reads_from_ascii(dk_engine *e, const uint8_t *seq, const uint64_t *offsets,
                              uint64_t n_reads, dk_reads **out);
/* host buffers already in the packed format above (copied to the device) */
dk_status dk_reads_from_packed(dk_engine *e, const uint64_t *bases, const uint64_t *mask,
                               uint64_t n_bases, uint64_t n_reads, uint64_t n_windows, dk_reads **out);
/* Overlapped ingest: as dk_reads_from_packed, but the copies run on the engine's copy stream and the call returns at
 * once; whatever consumes the batch (dk_set_insert, dk_probe, dk_accum_add, ...) waits for them on the device, so batch
 * i + 1 crosses PCIe while batch i is being worked on.  bases / mask must stay valid and unchanged until dk_reads_wait
 * returns, and should be pinned host memory (dk_host_alloc): pageable memory makes the call block. */
dk_status dk_reads_from_packed_async(dk_engine *e, const uint64_t *bases, const uint64_t *mask,
                                     uint64_t n_bases, uint64_t n_reads, uint64_t n_windows, dk_reads **out);
dk_status dk_reads_wait(dk_reads *r);                           /* blocks until the upload of r has finished */
dk_status dk_host_alloc(uint64_t bytes, void **out);            /* pinned host memory for the async upload */
void      dk_host_free(void *p);
/* device buffers already in the packed format (borrowed, not copied, not freed) */
dk_status dk_reads_attach_device(dk_engine *e, const void *d_bases, const void *d_mask,
                                 uint64_t n_bases, uint64_t n_reads, uint64_t n_windows, dk_reads **out);
/* synthetic reads first_read .. first_read+n_reads of sample (0,1 = parents, 2 = child), generated on the GPU */
dk_status dk_reads_synth(dk_engine *e, const dk_synth_config *cfg, int32_t sample,
                         uint64_t first_read, uint64_t n_reads, dk_reads **out);
dk_status dk_reads_stats(const dk_reads *r, dk_stats *out);    /* n_reads, n_bases, n_windows */
/* copy the packed words back (bases: ceil(n_bases/32) words, mask: ceil(n_bases/64) words) */
dk_status dk_reads_download(const dk_reads *r, uint64_t *bases, uint64_t *mask);
/* kmer.rs stand-in (extraction / canonicalisation / hashing as an operation of its own): for every
 * position p of the packed stream (read i starts at offsets[i] + i) the canonical k-mer starting
 * there -- kmers_lo[p], kmers_hi[p] (k > 32), hashes[p] -- and bit p of not_kmer (MSB-first words like
 * the batch's mask) = 1 where no k-mer starts (a non-ACGT base in the window, the window runs past
 * the read, separator); those positions get zeros.  n_bases entries per array, ceil(n_bases/64) mask
 * words.  Each pointer may be host or device memory; kmers_hi, hashes and not_kmer may be NULL. */
dk_status dk_reads_kmers(dk_engine *e, const dk_reads *r, uint64_t *kmers_lo, uint64_t *kmers_hi,
                         uint64_t *hashes, uint64_t *not_kmer, dk_stats *stats);
void      dk_reads_destroy(dk_reads *r);
/* host-side packer (no GPU): returns n_bases; bases/mask sized as for dk_reads_download */
uint64_t  dk_pack_ascii_host(const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                             uint64_t *bases, uint64_t *mask);

/* ---- KmerSet (replaces: KmerSet in counter.rs -- insert / contains / union) ----------------- */
dk_status dk_set_create(dk_engine *e, dk_set **out);                 /* zeroed, library-owned */
dk_status dk_set_attach(dk_engine *e, void *d_filter, dk_set **out); /* caller-owned device memory of 2^n/8 bytes;
                                                                        DK_SET_EXACT: call dk_set_clear before the first insert */
dk_status dk_set_clear(dk_set *s);
dk_status dk_set_insert(dk_set *s, const dk_reads *r, dk_stats *stats);      /* KmerSet::insert over all windows */
dk_status dk_set_contains(dk_set *s, const uint64_t *kmers_lo, const uint64_t *kmers_hi /* NULL if k<=32 */,
                          uint64_t n, uint8_t *out);                         /* KmerSet::contains */
dk_status dk_set_device_ptr(dk_set *s, void **d_filter, uint64_t *n_bytes);
dk_status dk_set_download(dk_set *s, uint64_t *words);               /* 2^n/64 words */
dk_status dk_set_upload(dk_set *s, const uint64_t *words);
dk_status dk_set_popcount(dk_set *s, uint64_t *n_bits_set);          /* DK_SET_EXACT: the number of k-mers held */
/* on-disk parent filter (reuse the parents across children): 64-byte header with the geometry
 * (k, canonical, filter_log2_bits, n_hashes, seed) followed by the 2^n/8 filter bytes.  dk_set_load
 * refuses a file whose geometry differs from the engine's. */
dk_status dk_set_save(dk_set *s, const char *path);
dk_status dk_set_load(dk_set *s, const char *path);
/* dst[i] |= src[j*slice_words + i] for j < n_slices: the local step of the OR-all-reduce
 * (RCCL has no bitwise-OR reduction; the host composes all-to-all -> this -> all-gather) */
dk_status dk_or_reduce_slices(dk_engine *e, void *d_dst, const void *d_src,
                              uint64_t n_slices, uint64_t slice_bytes);
/* DK_SET_EXACT counterpart: d_dst holds the 64-KiB segments first_segment .. first_segment +
 * slice_bytes/65536 of an exact set's table; the keys of the same segments in each of the n_slices
 * slices at d_src are inserted into it (the local step of the union-all-reduce: all-to-all -> this
 * -> all-gather).  DK_ERR_SET_FULL if a segment cannot take all keys. */
dk_status dk_union_slices(dk_engine *e, void *d_dst, const void *d_src, uint64_t n_slices,
                          uint64_t slice_bytes, uint64_t first_segment);
void      dk_set_destroy(dk_set *s);

/* ---- multi-GPU: one process (or thread) per GPU, one engine each; the parent set is the only thing exchanged ----
 * Reads shard across the ranks; every rank inserts its parent shard into a set of the FULL size; dk_set_allreduce_or
 * then combines the ranks' sets in place -- bitwise OR for a Bloom filter, key union for DK_SET_EXACT -- after which
 * every rank holds the whole parent set and probes its child shard locally.  RCCL has no OR reduction, so the
 * library composes it on the engine's stream: all-to-all of slices (ncclSend/ncclRecv) -> local OR / union kernel ->
 * ncclAllGather, in pieces that bound the staging memory (1 GiB).  librccl.so.1 is loaded on first use; the environment
 * variable DK_RCCL_LIBRARY names another file to load instead (no fallback if it does not load).
 *   dk_comm_unique_id  one rank creates the id; the host hands the 128 bytes to the other ranks (MPI, TCP, a file)
 *   dk_comm_init       collective over all ranks: joins the engine to the communicator (one per engine).  world_size 1
 *                      with id NULL needs no RCCL at all; world_size 1 WITH an id makes a real communicator of one rank
 *   dk_set_allreduce_or  collective; a no-op without a communicator (or with the RCCL-less one-rank kind); a one-rank
 *                      RCCL communicator runs the whole composition with the rank as its own peer -- the set is
 *                      unchanged (x | x = x) -- which lets a single GPU exercise the path.  bytes_sent may be NULL */
#define DK_COMM_ID_BYTES 128
dk_status dk_comm_unique_id(uint8_t *id /* DK_COMM_ID_BYTES */);
dk_status dk_comm_init(dk_engine *e, const uint8_t *id /* DK_COMM_ID_BYTES */, uint32_t rank, uint32_t world_size);
dk_status dk_comm_finalize(dk_engine *e);
dk_status dk_set_allreduce_or(dk_set *s, uint64_t *bytes_sent);
/* host-only arithmetic of the piece-wise exchanges (no GPU needed; tests): with `staging_bytes` of staging, slices of
 * slice_bytes and pieces that are multiples of `granule`, the bytes one piece carries and, for every peer q of `rank`,
 * the index of q's piece in the staging buffer (~0 for the rank itself when world_size > 1) */
dk_status dk_comm_layout(uint64_t staging_bytes, uint64_t slice_bytes, uint32_t rank, uint32_t world_size, uint64_t granule,
                         uint64_t *piece_bytes, uint64_t *staging_slot /* world_size entries */);

/* ---- membership pass + KmerCounter (replaces: child loop of counter.rs) ---------------------- */
/* probe every window of r against s; absent k-mers are counted.  s == NULL counts every k-mer
 * of the batch (KmerCounter semantics). */
dk_status dk_probe(dk_engine *e, dk_set *s, const dk_reads *r, dk_result **out, dk_stats *stats);
dk_status dk_result_size(const dk_result *res, uint64_t *n);
/* unordered dense copy; the destinations may be host or device memory; kmers_hi may be NULL when k <= 32 */
dk_status dk_result_copy(const dk_result *res, uint64_t *kmers_lo, uint64_t *kmers_hi, uint32_t *counts);
/* wrap caller-owned device arrays of n (k-mer, count) entries as a table (borrowed, not freed): the
 * receiving side of a multi-GPU merge -- gather the ranks' tables with RCCL, attach, dk_result_merge */
dk_status dk_result_attach(dk_engine *e, const void *d_kmers_lo, const void *d_kmers_hi /* NULL if k<=32 */,
                           const void *d_counts, uint64_t n, dk_result **out);
dk_status dk_result_device_view(const dk_result *res, const void **d_kmers_lo, const void **d_kmers_hi,
                                const void **d_counts, uint64_t *n);
/* Sum n_results tables by k-mer (a child processed in several batches, or the per-GPU tables of a
 * sharded run) and keep the k-mers with a summed count >= min_count.  The inputs should have been
 * produced with min_count = 1, otherwise k-mers below the threshold in every batch are already gone. */
dk_status dk_result_merge(dk_engine *e, const dk_result *const *results, uint32_t n_results,
                          uint32_t min_count, dk_result **out, dk_stats *stats);
void      dk_result_destroy(dk_result *res);

/* ---- child-only accumulator: a sample that arrives in many batches (replaces: the KmerCounter that the child
 * loop of counter.rs keeps across the whole BAM) -----------------------------------------------------------------
 * dk_probe returns the table of ONE batch; summing per-batch tables does not scale to a whole genome (a 30x child
 * leaves ~2 x 10^10 absent k-mer occurrences, nearly all distinct).  An accumulator keeps the absent occurrences
 * themselves -- 6- or 8-byte bucket records (16 for k > 32) grouped by hash prefix -- across any number of dk_accum_add
 * calls and counts each group once in dk_accum_finish, so counts and the min_count threshold are exact over the
 * whole sample.  When the records of the whole hash space do not fit beside the set, the sample is streamed
 * window_count times (a power of two): pass w keeps only the k-mers whose hash falls into the w-th of window_count
 * equal ranges; every pass touches 1/window_count of the set and of the partition workspace, and the union of the
 * passes' tables is the whole result (the ranges are disjoint).
 *   capacity_records  expected number of absent occurrences per pass (slack for the spread between groups is added
 *                     inside: the store takes ~1.1 x record bytes x capacity -- 6 bytes where the groups' common hash
 *                     prefix covers 16 bits or more, i.e. sets from 2^28 bits up, see dk_accum_geometry); occurrences beyond a group's room go to
 *                     an overflow list of capacity/64 entries, and DK_ERR_OVERFLOW is returned once that is full
 *                     (the accumulator is then unusable until dk_accum_reset; use more windows or a larger capacity)
 *   s == NULL         every k-mer counts (KmerCounter over a sample in batches) */
dk_status dk_accum_create(dk_engine *e, dk_set *s, uint32_t window_index, uint32_t window_count,
                          uint64_t capacity_records, dk_accum **out);
/* probe batch r against the set and append its absent occurrences inside the window; stats: n_reads, n_bases,
 * n_windows, n_valid (all valid windows of the batch) and n_absent (appended by this call) */
dk_status dk_accum_add(dk_accum *a, const dk_reads *r, dk_stats *stats);
/* (k-mer, count) of everything accumulated, counts >= min_count; stats: totals over all batches + n_distinct,
 * n_emitted.  The accumulator keeps its content (finish may be called again with another threshold). */
dk_status dk_accum_finish(dk_accum *a, uint32_t min_count, dk_result **out, dk_stats *stats);
/* empty the accumulator and move it to another window (same window_count): the next pass over the sample */
dk_status dk_accum_reset(dk_accum *a, uint32_t window_index);
dk_status dk_accum_stats(const dk_accum *a, dk_stats *out);     /* totals so far */
/* Multi-GPU (reads sharded over ranks): every rank accumulates its child shard over the same window, then the ranks
 * exchange unit ranges so that rank r holds units [r * n_units / P, (r + 1) * n_units / P) of every rank and counts
 * them -- the result stays sharded by hash range (the ranks' tables are disjoint; their union is the whole answer),
 * so counts and min_count are exact across the read shards and no rank ever holds the whole table.
 *   dk_accum_geometry      units of the window, records per unit, bytes per record: 16 (k > 32), 8, or 6 -- packed units, where all
 *                          records of a unit share at least 16 leading hash bits: a unit is a row of 384-byte blocks of 64 records,
 *                          64 x u32 (hash bits 0..31) then 64 x u16 (bits 32..47); the unit index supplies the bits above
 *   dk_accum_device_view   the store (n_units * unit_cap records, unit-major), the fills (n_units u32, records held by
 *                          each unit; values above unit_cap mean unit_cap) and the overflow list; synchronises
 *   dk_accum_finish_pieces counts units [first_unit, first_unit + n_units) from n_pieces (<= 8) slices laid out piece
 *                          after piece (d_stores: n_pieces * n_units * unit_cap records, d_fills: n_pieces * n_units
 *                          u32) -- what an all-to-all of the ranks' stores delivers -- plus d_extra, the all-gathered
 *                          overflow lists (records outside the unit range are skipped).  stats: n_absent = records
 *                          counted, n_distinct, n_emitted.  denovo_kmer_amd/dist.py:accum_exchange_finish is the recipe. */
dk_status dk_accum_geometry(const dk_accum *a, uint64_t *n_units, uint32_t *unit_cap, uint32_t *record_bytes);
dk_status dk_accum_device_view(dk_accum *a, void **d_store, void **d_fill, void **d_overflow, uint64_t *n_overflow);
dk_status dk_accum_finish_pieces(dk_accum *a, const void *d_stores, const void *d_fills, uint32_t n_pieces,
                                 uint64_t first_unit, uint64_t n_units, const void *d_extra, uint64_t n_extra,
                                 uint32_t min_count, dk_result **out, dk_stats *stats);
/* The same exchange + counting behind the C ABI, on the engine's communicator (dk_comm_init): collective.  The ranks
 * first compare accumulator geometry and state (a rank whose accumulator failed makes EVERY rank return an error, no
 * rank is left waiting), then swap unit ranges IN PLACE, piece by piece through the communicator's staging buffer
 * (grouped ncclSend / ncclRecv; no second copy of the store), all-gather the overflow lists, and each rank counts its
 * share of the units from the P pieces.  Afterwards the accumulator is consumed: dk_accum_reset before reuse.  Without a
 * communicator (one rank) it is dk_accum_finish.  stats->n_absent = occurrences this rank counted.  bytes_sent may be NULL. */
dk_status dk_accum_exchange_finish(dk_accum *a, uint32_t min_count, dk_result **out, dk_stats *stats, uint64_t *bytes_sent);
/* bytes of device memory the accumulator holds */
dk_status dk_accum_device_bytes(const dk_accum *a, uint64_t *n_bytes);
void      dk_accum_destroy(dk_accum *a);

#ifdef __cplusplus
}
#endif
#endif /* DENOVO_KMER_H */
