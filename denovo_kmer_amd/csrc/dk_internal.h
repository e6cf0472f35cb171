// dk_internal.h -- host-side state behind the opaque handles of include/denovo_kmer.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/denovo_kmer.h"
#include "dk_kernels_direct.h"

struct dk_pool_block {
    void *ptr;
    size_t bytes;
    bool in_use;
};

// An arena (dk_engine_reserve): one hipMalloc made ahead of time, carved up first by best fit; neighbouring free
// segments coalesce, so a workspace of any shape finds room as long as the bytes are there.
struct dk_arena_seg {
    size_t off, bytes;
    bool in_use;
};
struct dk_arena {
    char *base;
    size_t bytes;
    std::vector<dk_arena_seg> segs;       // sorted by offset, covering [0, bytes)
};

// what the last bucketed operation planned (dk_engine_get_info)
struct dk_plan_info {
    int levels = 0, b1 = 0, b2 = 0, b3 = 0, sbits = 0, slabs = 0, scan_variant = 0, T = 0;
};

// Run-time options of an engine (dk_engine_set_option): the capacity hint is for callers, the rest are the
// test hooks that force a kernel geometry on inputs too small to select it (validated; 0 = automatic).
struct dk_options {
    int multiplicity_hint = 0;    // expected copies of one k-mer inside one batch (0 = assume up to 64)
    int scan_variant = 0;         // scan_part tile geometry 1..6
    int repart_variant = 0;       // 1 = 1024 x 16 repart tiles
    int force_l3 = 0;             // three partition levels from 8 segments on
    int b1_up = 0;                // shift the level-1 / level-2 bit split
    int count_seg = 0;            // KmerCounter: records per counting segment (default 5000)
    int cnt_mid = 0;              // seg_count: threshold of the 512-thread geometry (default 3900)
    int cnt_big = 0;              // seg_count: threshold of the 1024-thread geometry (default 7900, k > 32: 3500)
    int cnt_split_to = 0;         // absent-list split: records per unit aimed at (default 6000, k > 32: 3000)
    int repart_bits = 0;          // repart: most hash bits one pass may take (default 10; 9 = round 1's limit, for A/B runs)
    int slabs = 0;                // slab-wise level 2 of insert / accumulate: number of slabs (0 = automatic; a power of two)
    int slab_mb = 0;              // automatic slabs: room for one slab's regions in MiB (default 12288)
    int ovf_cap = 0;              // capacity of the partition's overflow list in records (test hook; 0 = an eighth of the batch)
    int accum_min_u = 0;          // dk_accum_create: at least 2^n counting units per segment (test hook: packed units on small sets)
    int scan_positions = 0;       // scan_part: 1 = always position-major, never window-major (A/B runs, tests)
    int kmers_plain = 0;          // dk_reads_kmers: 1 = ordinary stores for the outputs instead of non-temporal ones (A/B)
    int mode = 0;                 // kernel family override: 0 = dk_config.mode, 1 = direct, 2 = bucketed
    int l2_packed = 0;            // 1 = level-2 regions of >= 16 prefix bits hold packed 6-byte records (slower: DESIGN.md section 9; A/B runs, tests)
    int accum_plain = 0;          // dk_accum_create: 1 = never use packed 6-byte unit records (A/B runs, tests)
    int scan_bits = 0;            // scan_part: most hash bits level 1 may take (default 10; 9 = round 2's limit, for A/B runs)
    int repart_pieces = 0;        // repart: 0 = automatic, 1 = one workgroup per tile of a PIECE (round 2), 2 = tiles over a bin's concatenated pieces
    int repart_plain = 0;         // repart: 1 = tiles in plain block order instead of one bin per XCD (A/B runs)
    int sub_split = 0;            // sub-segment split of the set kernels: 0 = automatic, 1..3 = force, 9 = never
    int merge_pass_bits = 0;      // dk_result_merge: at least 2^n hash-range passes
    int accum_unit_cap = 0;       // dk_accum_create: records per counting unit, when at least what the capacity needs (test hook)
    int sink_plain = 0;           // dk_probe: never sink the absent records into finer counting units (test hook)
    int l1_skew = 0;              // bytes between the level-1 pieces of consecutive bins (0 = 128; A/B runs)
    int l1_layout = 0;            // level-1 pieces: 0 = bin-major, 1 = workgroup-major (measured equal; A/B runs, tests)
    int comm_staging_kb = 0;      // dk_comm_init: size of the staging buffer in KiB (0 = 1 GiB; tests: many pieces on small sets)
    int merge_undersize = 0;      // dk_result_merge: start with pass tables 2^n times too small (test hook: the redo path)
    int merge_idx64 = 0;          // dk_result_merge: 64-bit candidate indices whatever the size
};

struct dk_comm;
constexpr int DK_MAX_MARKS = 320;

struct dk_engine {
    dk_config cfg;
    dk_options opt;
    dk_comm *comm = nullptr;      // RCCL communicator (dk_comm_init), or none
    int device;
    int n_cu;
    hipStream_t stream;
    bool own_stream;
    hipStream_t copy_stream = nullptr;    // uploads of dk_reads_from_packed_async (created on first use)
    hipEvent_t copy_ev = nullptr;         // orders an upload behind the work already queued on `stream`
    std::string err;
    dk::Counters *d_ctr;          // device counters of the running operation
    dk::Counters *h_ctr;          // pinned host mirror
    std::vector<dk_pool_block> pool;
    std::vector<dk_arena> arenas;
    dk_plan_info plan;
    uint64_t pool_peak = 0;       // most bytes ever handed out at one time
    // stage timing of the last operation
    // (a slab-wise operation marks two stages per slab: the marks are summed by name into the dk_timings entries)
    hipEvent_t ev[DK_MAX_MARKS + 1];
    int n_ev;
    char ev_name[DK_MAX_MARKS][24];
    dk_timings timings;
};

struct dk_reads {
    dk_engine *e;
    uint64_t *d_bases;
    uint64_t *d_mask;
    uint64_t n_bases, n_reads, n_windows;
    bool owns;
    hipEvent_t ready = nullptr;   // dk_reads_from_packed_async: the upload's completion on the copy stream
    // Uniform read length (window-major scan, dk_bucket_scan.h): stride = L + 1 when the batch may consist of reads of one
    // length L (0 = no); *d_uniform (device) = 1 once a kernel has verified that every position = L mod stride is flagged
    uint32_t stride = 0;
    uint32_t *d_uniform = nullptr;
};

struct dk_set {
    dk_engine *e;
    unsigned long long *d_words;
    uint64_t n_bytes;
    bool owns;
    bool exact;                   // DK_SET_EXACT: d_words is read as open-addressing tables (dk_device.h)
};

// Child-only accumulator (dk_accum_*): the absent k-mer occurrences of many batches, kept as bucket records
// grouped by counting unit (a hash-prefix range) and counted once by dk_accum_finish.
struct dk_accum {
    dk_engine *e;
    dk_set *s;                    // parent set (NULL: every k-mer counts, KmerCounter over many batches)
    int wbits;                    // the accumulator covers the hashes whose top wbits equal widx
    uint32_t widx;
    int T;                        // segment bits of the engine's set geometry (all windows together)
    int u;                        // counting units per segment = 2^u
    uint64_t n_units;             // 2^(T - wbits + u)
    uint32_t unit_cap;            // records per unit
    void *store;                  // n_units * unit_cap records (Rec1 / Rec2)
    uint32_t *fill;               // records held per unit
    void *ovf;                    // records that found their unit full (counted with the unit at the end)
    uint64_t ovf_cap;
    unsigned long long *d_novf;   // device counter of the overflow list
    uint64_t n_absent, n_valid, n_windows, n_reads, n_bases, n_batches;
    bool wide;
    bool packed;                  // units hold 6-byte packed records (k <= 32 and T + u >= 16; dk_bucket_seg.h)
    bool exchanged;               // dk_accum_exchange_finish transposed the store in place: reset before reuse
    bool failed;                  // a batch lost records (overflow list full) or died half-way: reset before reuse
};

struct dk_result {
    dk_engine *e;
    uint64_t *d_lo, *d_hi;
    uint32_t *d_cnt;
    uint64_t n;
    bool wide;
    bool owns;                    // false: the arrays belong to the caller (dk_result_attach)
    // Region r of the arrays holds region_n[r] entries starting at r * region_cap.  The direct
    // family writes one dense region; the bucketed count kernel appends through RESULT_REGIONS
    // independent fill counters (one global counter saturates near 10^8 atomics/s) and the regions
    // are stitched on copy-out, or compacted on the first dk_result_device_view.
    uint32_t n_regions;
    uint64_t region_cap;
    uint64_t region_n[dk::RESULT_REGIONS];
};

namespace dk {

dk_status fail(dk_engine *e, dk_status s, const char *fmt, ...);
dk_status pool_alloc(dk_engine *e, size_t bytes, void **out);
void pool_free(dk_engine *e, void *p);
void stage_begin(dk_engine *e);
void stage_mark(dk_engine *e, const char *name);
dk_status stage_end(dk_engine *e);      // synchronises the stream and fills e->timings
int grid_for(const dk_engine *e, uint64_t n_threads, int block);

}  // namespace dk

#define DK_HIP(e, call)                                                                      \
    do {                                                                                     \
        hipError_t _r = (call);                                                              \
        if (_r != hipSuccess)                                                                \
            return dk::fail((e), _r == hipErrorOutOfMemory ? DK_ERR_OOM : DK_ERR_HIP,        \
                            "%s failed: %s (%s:%d)", #call, hipGetErrorString(_r), __FILE__, \
                            __LINE__);                                                       \
    } while (0)

#define DK_TRY(call)                         \
    do {                                     \
        dk_status _s = (call);               \
        if (_s != DK_OK) return _s;          \
    } while (0)
