// dk_bucket_host.h -- host side of the bucketed family: plans, partition, the insert / probe / count / accumulate stages
// (part of the bucketed kernel family: dk_kernels_bucket.h has the overview and includes the parts in order)
#pragma once
#include "dk_bucket_scan.h"
#include "dk_bucket_repart.h"
#include "dk_bucket_seg.h"
#include "dk_bucket_count.h"
#include "dk_bucket_rare.h"

namespace dk {

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

// most hash bits a multisplit level may take: level 1 (scan_part, 1024 threads: 10; option "scan_bits" = 9 restores round 2's
// limit for A/B runs) and later levels (repart: 10, option "repart_bits"); k > 32 runs 512-thread kernels: 9 each
inline int level1_bits(const dk_engine *e) { return e->cfg.k > 32 ? MAX_BIN_BITS : (e->opt.scan_bits > 0 ? e->opt.scan_bits : MAX_BIN_BITS1); }
inline int level2_bits(const dk_engine *e) { return e->cfg.k > 32 ? MAX_BIN_BITS : (e->opt.repart_bits > 0 ? e->opt.repart_bits : MAX_BIN_BITS2); }

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
    const int two = level1_bits(e) + level2_bits(e);
    return T_local == two + 1 ? 1 : 0;
}

// T_override > 0: number of segment bits to use instead of the filter's.  wbits > 0: only the records of one
// hash window (1 / 2^wbits of them) are partitioned, over the T - wbits segment bits below the window's.
// sbits > 0 (insert / accumulate against a set): the partition stops sbits bits short of the 64-KiB segments, the segment
// kernels resolve them (PieceList::sbits); p->T, n_seg, cap2 then describe the regions.
inline bool make_plan(const dk_engine *e, const dk_reads *r, BucketPlan *p, int T_override = 0, int wbits = 0, int sbits = 0,
                      bool allow_slabs = false)
{
    const bool wide = e->cfg.k > 32;
    p->sbits = sbits;
    p->slabs = 1;
    p->packed2 = false;
    p->T = (T_override > 0 ? T_override : set_segment_bits(e)) - wbits - sbits;
    if (p->T < 1 || p->T > MAX_SEG_BITS) return false;
    p->b3 = 0;
    p->capA = 0;
    const int bits1 = level1_bits(e), bits2 = level2_bits(e);
    if (p->T > bits1 + bits2 || (e->opt.force_l3 && p->T >= 3)) {
        // three levels: thirds of T; the coarse regions (b1 + b2 bits) index the grid's y dimension
        p->b1 = p->T / 3;
        p->b2 = (p->T - p->b1) / 2;
        p->b3 = p->T - p->b1 - p->b2;
    } else {
        p->b1 = (p->T + e->opt.b1_up) / 2;
        if (p->b1 > bits1) p->b1 = bits1;
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
    // (a tile of positions holds n_windows / n_bases windows on average: 120 / 151 for 150-bp reads and k = 31 -- sizing
    // the pieces by positions cost 26 % more level-1 room than the records need; batches whose read lengths differ a lot
    // between tiles lean on the 8 sigma and on the overflow list)
    const double win_frac = r->n_bases ? std::min(1.0, 1.02 * (double)n_all / (double)r->n_bases) : 1.0;
    const double share1 = std::min((double)p->n_max, (double)(tiles_per_wg * (uint64_t)p->tile) * win_frac / (double)(1ULL << wbits));
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
    // slabs: a power of two, at least 8 level-1 bins each (one per XCD for repart); automatic = the fewest that keep one
    // slab's regions within the budget (option "slab_mb")
    if (allow_slabs && !p->b3 && p->p1 >= 16) {
        const uint32_t max_slabs = p->p1 / 8;
        uint32_t S = 1;
        if (e->opt.slabs > 0) {
            while (S * 2 <= (uint32_t)e->opt.slabs && S * 2 <= max_slabs) S *= 2;
        } else {
            // (fewer, larger slabs are faster -- every launch ends with a tail of half-empty CUs: 8 slabs 83.3 Gk-mers/s,
            // 64 slabs 79.5, 128 slabs 77.5 on the whole-genome child step -- and nothing of a slab survives in the
            // Infinity Cache between its two kernels anyway; 12 GiB keeps the whole-genome child at 8 slabs)
            const uint64_t budget = (uint64_t)(e->opt.slab_mb > 0 ? e->opt.slab_mb : 12288) << 20;
            const uint64_t rec_bytes = wide ? 16 : 8;
            while (S < max_slabs && p->n_seg * (uint64_t)p->cap2 * rec_bytes / S > budget) S *= 2;
        }
        p->slabs = S;
    }
    p->slab_bins = p->p1 / p->slabs;
    // (the flows that take slabs -- insert and accumulate -- are the ones whose set kernels read packed regions)
    // Off unless option "l2_packed" asks for it: the regions' 16 fewer bits per record save 12 % of repart's writes and 25 % of
    // the set kernels' record reads, but a record then takes TWO store instructions in repart's copy-out (u32 + u16), and that
    // kernel is bound by its store issue rate: whole-genome child step repart 21.4 -> 27.2 ms against seg_probe 22.3 -> 22.6
    // (no gain: not bound by its record reads); parent batch repart 50.7 -> 55.9, seg_insert 51.7 -> 46.0 ms.
    p->packed2 = allow_slabs && !wide && !p->b3 && e->opt.l2_packed && wbits + p->T >= PACKED_MIN_PREFIX_BITS;
    if (p->packed2) {
        p->cap2 = (p->cap2 + 63) / 64 * 64;                      // whole 64-record blocks
        if (((uint64_t)p->cap2 * PACKED_REC_BYTES) % 16384 == 0) p->cap2 += 64;
    }
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

// level 2 of one slab (p.slabs == 1: of the whole batch): the level-1 pieces of the slab's bins -> the regions of these
// Where the level-1 pieces lie in B.a: piece w of bin b starts at record b * bin_stride + w * piece_stride.
//   bin-major (default): a bin's G pieces side by side, 128 bytes between bins
//   workgroup-major (option "l1_layout" = 1): a scan_part workgroup's 2^b1 pieces side by side, 128 bytes between workgroups --
//   its 2^b1 write frontiers then lie within 2^b1 * capw records (180 MB at configs[2]: ~90 pages of 2 MiB) instead of one per
//   bin across the whole 46-GB buffer.  Measured equal (child step: scan_part 20.8 both, repart 20.4 vs 20.3; parent batch:
//   scan_part 50.1 vs 51.8, repart 51.5 vs 51.0): page reach is not what holds the scan's stores.
struct Level1Layout {
    uint64_t bin_stride, piece_stride, total;
};
template <class R>
inline Level1Layout level1_layout(const dk_engine *e, const BucketPlan &p)
{
    const uint64_t skew = (uint64_t)(e->opt.l1_skew > 0 ? e->opt.l1_skew : 128) / (uint32_t)sizeof(R);      // (option "l1_skew": bytes, a multiple of 16)
    if (e->opt.l1_layout != 1) return Level1Layout{(uint64_t)p.G * p.capw + skew, p.capw, (uint64_t)p.p1 * ((uint64_t)p.G * p.capw + skew)};
    return Level1Layout{p.capw, (uint64_t)p.p1 * p.capw + skew, (uint64_t)p.G * ((uint64_t)p.p1 * p.capw + skew)};
}

// bins in B.b (two levels), or -> 2^(b1+b2) coarse regions (three levels; never slab-wise)
template <bool WIDE>
inline void launch_repart(dk_engine *e, const BucketPlan &p, BucketBufs<typename RecOf<WIDE>::type> &B, int wbits, uint32_t slab)
{
    using R = typename RecOf<WIDE>::type;
    const OvfList<R> ovf{B.ovf, &e->d_ctr->n_ovf, B.ovf_cap};
    const Level1Layout L1 = level1_layout<R>(e, p);
    const uint32_t bin0 = slab * p.slab_bins;
    const int affine = !e->opt.repart_plain && p.slab_bins % 8 == 0;
#define DK_REPART_LAUNCH(TH, PT, W)                                                                       \
    do {                                                                                                  \
        const uint32_t tpp = (p.capw + TH * PT - 1) / (TH * PT);                                           \
        repart_kernel<TH, PT, W, R><<<repart_grid(p.G * tpp, p.slab_bins), TH, 0, e->stream>>>(            \
            B.a, B.cnt1, p.G, p.capw, tpp, wbits + p.b1, p.b2, p.b3 ? p.capA : p.cap2, B.b,                 \
            p.b3 ? B.cursorA : B.cursor2, ovf, e->d_ctr, affine, L1.bin_stride, L1.piece_stride, bin0, slab);                      \
    } while (0)
    // The pieces of a bin read as one array (full tiles) where tiles cut piece by piece would be poorly filled: the
    // concatenating kernel costs ~6 % more per tile (prefix sum of the piece sizes, bisection), so it is taken when the
    // piece-wise launch would provide more than 1.25 times the tile capacity the records need (whole-genome child step,
    // 22 K-record pieces in four 8 K tiles: repart 21.6 -> 20.4 ms; parent batch, 59 K-record pieces in eight: 50.7 -> 54.0,
    // so it stays piece-wise).  Option "repart_pieces": 1 = always piece-wise, 2 = always concatenated.
    const uint64_t mean_bin = p.n_max / p.p1;
    const uint32_t tile_recs = WIDE ? 512 * 8 : 1024 * 8;
    const uint64_t launched = (uint64_t)((p.capw + tile_recs - 1) / tile_recs) * tile_recs * p.G;
    const bool concat = p.G >= 2 && p.G <= 512 && e->opt.repart_pieces != 1 &&
                        (e->opt.repart_pieces == 2 || launched * 4 > mean_bin * 5);
#define DK_REPART_CONCAT(TH, PT, W)                                                                       \
    do {                                                                                                  \
        const uint32_t tpb = (uint32_t)(((uint64_t)p.G * p.capw + TH * PT - 1) / (TH * PT));               \
        repart_kernel<TH, PT, W, R, false, true><<<repart_grid(tpb, p.slab_bins), TH, 0, e->stream>>>(     \
            B.a, B.cnt1, p.G, p.capw, tpb, wbits + p.b1, p.b2, p.b3 ? p.capA : p.cap2, B.b,                 \
            p.b3 ? B.cursorA : B.cursor2, ovf, e->d_ctr, affine, L1.bin_stride, L1.piece_stride, bin0, slab);                      \
    } while (0)
    if (concat && !p.packed2 && !(e->opt.repart_variant == 1 && !WIDE)) {
        if constexpr (WIDE) DK_REPART_CONCAT(512, 8, 8);
        else DK_REPART_CONCAT(1024, 8, 8);
    } else if constexpr (WIDE) DK_REPART_LAUNCH(512, 8, 8);
    else if (p.packed2) {
        const uint32_t tpp = (p.capw + 1024 * 8 - 1) / (1024 * 8);
        repart_kernel<1024, 8, 8, R, true><<<repart_grid(p.G * tpp, p.slab_bins), 1024, 0, e->stream>>>(
            B.a, B.cnt1, p.G, p.capw, tpp, wbits + p.b1, p.b2, p.cap2, B.b, B.cursor2, ovf, e->d_ctr, affine, L1.bin_stride, L1.piece_stride, bin0, slab);
    } else if (e->opt.repart_variant == 1) DK_REPART_LAUNCH(1024, 16, 4);
    else DK_REPART_LAUNCH(1024, 8, 8);
#undef DK_REPART_LAUNCH
#undef DK_REPART_CONCAT
}

// scan_part + repart (+ repart): afterwards B.rec / B.cursor2 hold every record of the batch (of the hash window
// widx of 2^wbits, when wbits > 0) grouped by segment, except the records that did not fit, which are in B.ovf
// (Counters::n_ovf of them).  need_scratch: a second segment-sized buffer for the absent lists (per-batch probe).
// Slab-wise plans (p.slabs > 1): only scan_part runs here and B.b is sized for ONE slab; the caller walks the slabs
// (launch_repart + its set kernel, see slab_piece_list).
template <bool WIDE>
inline dk_status bucketed_partition(dk_engine *e, const dk_reads *r, const BucketPlan &p,
                                    BucketBufs<typename RecOf<WIDE>::type> &B, int wbits = 0, uint32_t widx = 0,
                                    bool need_scratch = true)
{
    using R = typename RecOf<WIDE>::type;
    e->plan = dk_plan_info{p.b3 ? 3 : 2, p.b1, p.b2, p.b3, p.sbits, (int)p.slabs, p.variant, p.T};
    const uint64_t seg_recs = p.n_seg / p.slabs * (uint64_t)p.cap2;
    // 128 bytes between the pieces of consecutive level-1 bins: a workgroup of scan_part writes to 2^b1 frontiers that are
    // G * capw records apart, always a multiple of 4 KiB, so all of them sat on the same few HBM channels at any moment
    // (configs[1]: scan_part 4.75 -> 4.35 ms on one box, no difference on others)
    const Level1Layout L1 = level1_layout<R>(e, p);
    const uint64_t lvl1_recs = L1.total;
    const uint64_t n_coarse = p.b3 ? 1ULL << (p.b1 + p.b2) : 0;
    const uint64_t coarse_recs = n_coarse * p.capA;
    // two levels: a = level-1 pieces (then the absent lists), b = segments.  three: a = level 1, then segments; b = coarse (then absent lists)
    const uint64_t a_recs = p.b3 ? std::max(seg_recs, lvl1_recs) : std::max(need_scratch ? seg_recs : 0, lvl1_recs);
    const uint64_t b_recs = p.b3 ? std::max(need_scratch ? seg_recs : 0, coarse_recs) : seg_recs;
    DK_TRY(pool_alloc(e, a_recs * sizeof(R), (void **)&B.a));
    DK_TRY(pool_alloc(e, p.packed2 ? seg_recs * PACKED_REC_BYTES : b_recs * sizeof(R), (void **)&B.b));
    const uint64_t n1 = (uint64_t)p.p1 * p.G;
    DK_TRY(pool_alloc(e, (n1 + 2 * p.n_seg + n_coarse) * 4, (void **)&B.cnt));
    B.cnt1 = B.cnt;
    B.cursor2 = B.cnt + n1;
    B.miss_cnt = B.cursor2 + p.n_seg;
    B.cursorA = B.miss_cnt + p.n_seg;
    B.rec = p.b3 ? B.a : B.b;
    B.scratch = p.b3 ? B.b : B.a;
    if (n_coarse) DK_HIP(e, hipMemsetAsync(B.cursorA, 0, n_coarse * 4, e->stream));
    B.ovf_cap = e->opt.ovf_cap > 0 ? (uint64_t)e->opt.ovf_cap : std::max<uint64_t>(1ULL << 20, p.n_max / (p.n_max >> 30 ? 16 : 8));
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
    // window-major thread mapping for batches of one read length (k <= 32; the kernel looks at the batch's verified flag)
    WindowMajor wmv{0, 0, 0, 0, nullptr};
    if (!WIDE && !e->opt.scan_positions && r->stride > e->cfg.k && r->d_uniform) {
        const uint32_t W = r->stride - e->cfg.k;                         // windows per read: L - k + 1 = stride - k
        const uint32_t tpr = (W + 15) / 16, wpt = (W + tpr - 1) / tpr;
        if ((uint64_t)r->n_reads * tpr < 0x7FFFFFFFULL) wmv = WindowMajor{r->stride, tpr, wpt, (uint32_t)(r->n_reads * tpr), r->d_uniform};
    }
#define DK_SCAN_LAUNCH(TH, PT, W, WIN)                                                                                    \
    scan_part_kernel<TH, PT, W, WIDE, WIN><<<p.G, TH, 0, e->stream>>>(sv, (int)e->cfg.k, (int)e->cfg.canonical,           \
                                                                      e->cfg.seed, p.b1, p.capw, B.a, B.cnt1, n_tiles,    \
                                                                      ovf, e->d_ctr, wbits, widx, L1.bin_stride, L1.piece_stride, \
                                                                      (PT == 16 ? wmv : WindowMajor{0, 0, 0, 0, nullptr}))
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
    }
    DK_HIP(e, hipGetLastError());
    stage_mark(e, "scan_part");
    if (p.slabs > 1) return DK_OK;                          // the caller walks the slabs
    launch_repart<WIDE>(e, p, B, wbits, 0);
    if (p.b3) {
        DK_HIP(e, hipGetLastError());
        stage_mark(e, "repart");
        if constexpr (WIDE) DK_REPART3_LAUNCH(512, 8, 8);
        else DK_REPART3_LAUNCH(1024, 8, 8);
    }
#undef DK_SCAN_LAUNCH
#undef DK_REPART3_LAUNCH
    DK_HIP(e, hipGetLastError());
    stage_mark(e, p.b3 ? "repart3" : "repart");
    return DK_OK;
}

// the piece list of slab `slab`'s regions (B.b holds one slab at a time; the region counts are indexed globally)
template <class R>
inline PieceList<R> slab_piece_list(const BucketPlan &p, const BucketBufs<R> &B, uint32_t slab, int wbits = 0, uint32_t widx = 0)
{
    const uint64_t regions_per_slab = p.n_seg / p.slabs;
    PieceList<R> pl{B.rec, B.cursor2 + slab * regions_per_slab, 1, p.cap2, nullptr, nullptr};
    pl.pk_bits = wbits + p.T;                                    // (read by the packed kernels only)
    pl.pk_region0 = ((uint64_t)widx << p.T) + slab * regions_per_slab;
    return pl;
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
    if (e->h_ctr->n_overflow || e->h_ctr->n_sink_drop)
        return fail(e, DK_ERR_OVERFLOW, "bucket overflow (%llu records)",
                    (unsigned long long)(e->h_ctr->n_overflow + e->h_ctr->n_sink_drop));
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
    if (!make_plan(e, r, &p, 0, 0, pick_sub_bits(e, T_full), true)) return fail(e, DK_ERR_UNSUPPORTED, "no bucketed plan for this geometry");
    BucketBufs<R> B;
    dk_status st = bucketed_partition<WIDE>(e, r, p, B, 0, 0, false);
    if (st == DK_OK) {
        // slab by slab (one slab = everything, unless the plan is slab-wise): level 2 of the slab's bins, then its segments
        const uint64_t regions_per_slab = p.n_seg / p.slabs;
        const unsigned n_seg = (unsigned)(regions_per_slab << p.sbits);
        hipError_t h = hipSuccess;
        for (uint32_t sl = 0; sl < p.slabs && h == hipSuccess; sl++) {
            if (p.slabs > 1) {
                launch_repart<WIDE>(e, p, B, 0, sl);
                stage_mark(e, "repart");
            }
            PieceList<R> pl = slab_piece_list(p, B, sl);
            pl.sbits = p.sbits;
            pl.sub_shift = 64 - T_full;
            const uint64_t seg_base = (uint64_t)sl * n_seg;
            bool launched = false;
            if constexpr (!WIDE) {
                if (p.packed2) {
                    if (s->exact)
                        seg_exact_insert_kernel<R, true><<<n_seg, SEG_THREADS, 0, e->stream>>>(s->d_words, pl, T_full, e->d_ctr, seg_base);
                    else
                        seg_insert_kernel<R, true><<<n_seg, SEG_THREADS, 0, e->stream>>>(
                            s->d_words, pl, (int)e->cfg.n_hashes, 64 - T_full - SEG_LOG2_BLOCKS, seg_base);
                    launched = true;
                }
            }
            if (launched) {
            } else if (s->exact)
                seg_exact_insert_kernel<R><<<n_seg, SEG_THREADS, 0, e->stream>>>(s->d_words, pl, T_full, e->d_ctr, seg_base);
            else
                seg_insert_kernel<R><<<n_seg, SEG_THREADS, 0, e->stream>>>(
                    s->d_words, pl, (int)e->cfg.n_hashes, 64 - T_full - SEG_LOG2_BLOCKS, seg_base);
            h = hipGetLastError();
            if (h == hipSuccess) stage_mark(e, s->exact ? "seg_exact_insert" : "seg_insert");
        }
        if (h == hipSuccess) {
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
template <class R, int ACC, bool PK = false>
inline hipError_t launch_seg_probe(dk_engine *e, dk_set *s, const PieceList<R> &list, uint64_t n_seg, int T_full,
                                   uint64_t seg_base, const MissOut<R> &mo)
{
    const int blk_shift = 64 - T_full - SEG_LOG2_BLOCKS;
    if (s && s->exact)
        seg_exact_probe_kernel<R, ACC, PK><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(s->d_words, list, T_full, seg_base, mo, e->d_ctr);
    else if (s && e->cfg.n_hashes == 4)
        seg_probe_kernel<R, 4, ACC, PK><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(s->d_words, list, 4, blk_shift, seg_base, mo, e->d_ctr);
    else
        seg_probe_kernel<R, 0, ACC, PK><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(
            s ? s->d_words : nullptr, list, s ? (int)e->cfg.n_hashes : 0, blk_shift, seg_base, mo, e->d_ctr);
    return hipGetLastError();
}

// Count the records of `list` unit by unit into res (seg_count): n_units units whose hashes share the top Tc bits
// (unit_base + local index), n_absent records in all, extra_room more entries per region.
// The table is sized for every record being distinct when min_count == 1.  With min_count > 1 few records
// survive (a whole-genome child keeps ~1.5 % of its absent occurrences at min_count 2): the table is then sized for
// a sixteenth of the upper bound n_absent / min_count, and if a region runs out the kernel has still tallied what
// each region needs (region_fill), so the count is redone once with exactly that much room.
template <bool WIDE>
inline dk_status bucketed_count_stage(dk_engine *e, const PieceList<typename RecOf<WIDE>::type> &list, uint64_t n_units,
                                      int Tc, uint64_t unit_base, uint64_t n_absent, uint64_t extra_room, uint32_t min_count,
                                      dk_result *res, uint64_t size_records = 0, bool packed = false)
{
    if (!n_absent) return DK_OK;
    const uint64_t per_seg = n_absent / n_units;
    // packed: `list` describes units of an accumulator's packed store (k <= 32 only)
#define DK_COUNT_LAUNCH(TH, SLOTS, BM, PER_CU)                                                                                  \
    do {                                                                                                                        \
        const unsigned cgrid = (unsigned)std::min<uint64_t>(n_units, (uint64_t)e->n_cu * PER_CU);                               \
        /* (k <= 32 has the LDS for bitmaps of twice the words below the 1024-thread geometry: >= 16 bits per record up to the */ \
        /* geometry's capacity -- 6 % instead of 11 % of the unique records take the table path) */                               \
        constexpr int BMW = (WIDE || TH >= 1024) ? BM : 2 * BM;                                                                  \
        if (packed && !WIDE && list.n_pieces == 1 && !list.extra)                                                              \
            seg_count_kernel<TH, SLOTS, BMW, WIDE, !WIDE, WIDE ? 0 : 1><<<cgrid, TH, 0, e->stream>>>(                           \
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);   \
        else if (packed && !WIDE && list.n_pieces == 1)                                                                         \
            seg_count_kernel<TH, SLOTS, BMW, WIDE, !WIDE, WIDE ? 0 : 2><<<cgrid, TH, 0, e->stream>>>(                           \
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);   \
        else if (packed && !WIDE)                                                                                               \
            seg_count_kernel<TH, SLOTS, BMW, WIDE, !WIDE><<<cgrid, TH, 0, e->stream>>>(                                         \
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);   \
        else                                                                                                                    \
            seg_count_kernel<TH, SLOTS, BMW, WIDE, false><<<cgrid, TH, 0, e->stream>>>(                                         \
                list, n_units, Tc, e->cfg.seed, min_count, region_cap, res->d_lo, res->d_hi, res->d_cnt, e->d_ctr, unit_base);   \
    } while (0)
    auto launch = [&](uint64_t region_cap) -> hipError_t {
        // Each geometry keeps a unit in registers up to threads x 16 records (k > 32: x 8), and at a given unit size the smaller
        // geometry wins (more workgroups per CU overlap their phases: 3.8 K-record units of the whole-genome steps 4.66 ms per step
        // with 256 threads, 6.74 with 512, 11.6 with 1024), so k <= 32 takes each geometry up to the mean that still leaves 3 sigma
        // below its capacity
        if (per_seg >= (e->opt.cnt_big > 0 ? (uint64_t)e->opt.cnt_big : (WIDE ? 3500u : 7900u))) {
            // big segments: 1024 threads hold 8K (k > 32) / 16K records in registers, 256-Kbit bitmaps
            DK_COUNT_LAUNCH(1024, 2048, 8192, 2);
        } else if (per_seg >= (WIDE ? 1300u : (uint64_t)(e->opt.cnt_mid > 0 ? e->opt.cnt_mid : 3900))) {
            DK_COUNT_LAUNCH(512, 2048, 2048, 6);
        } else if (per_seg >= (WIDE ? 600u : 1200u)) {
            // 256 threads hold 2K (k > 32) / 4K records: 2^17 segments at configs[1] leave ~1.6 K absent records each
            DK_COUNT_LAUNCH(256, 1024, 1024, 12);
        } else {
            DK_COUNT_LAUNCH(128, 512, 256, 32);
        }
        return hipGetLastError();
    };
#undef DK_COUNT_LAUNCH
    // RESULT_REGIONS output regions, each with its own fill counter; segments are dealt to the
    // regions round-robin, so the regions fill evenly (12.5 % + 64 Ki entries of slack each);
    // overflow records may all sit in one segment, hence the extra room for them
    const uint64_t used_regions = std::min<uint64_t>(RESULT_REGIONS, n_units);
    // size_records (accumulators: their capacity): the optimistic table is sized from it instead of from n_absent, so that
    // every counting pass of one accumulator asks the pool for the same block and none of them waits for hipMalloc
    // (a whole-genome child keeps 1.5 % of its absent occurrences at min_count 2: a sixteenth of the bound is 2.4 times that)
    const uint64_t bound = min_count > 1 ? std::max(n_absent, size_records) / min_count / 16 : n_absent;
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
            hipError_t h = launch_seg_probe<R, ACC_NONE>(e, s, list, n_sample, p.T, 0, mo);
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
                        if (h == hipSuccess) h = launch_seg_probe<R, ACC_PLAIN>(e, s, list, p.n_seg, p.T, 0, mf);
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
                            e->h_ctr->n_overflow = e->h_ctr->n_sink_drop = 0;
                            e->h_ctr->n_absent = 0;
                            snprintf(e->ev_name[e->n_ev - 1], sizeof e->ev_name[0], "seg_probe_sunk");   // the abandoned attempt
                            h = hipMemsetAsync(&e->d_ctr->n_overflow, 0, 8, e->stream);
                            if (h == hipSuccess) h = hipMemsetAsync(&e->d_ctr->n_sink_drop, 0, 8, e->stream);
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
            const hipError_t h = launch_seg_probe<R, ACC_NONE>(e, s, list, p.n_seg, p.T, 0, mo);
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
inline size_t accum_rec_bytes(const dk_accum *a) { return a->wide ? sizeof(Rec2) : a->packed ? (size_t)PACKED_REC_BYTES : sizeof(Rec1); }

// Partition the batch's records of the accumulator's hash window, test them against the set and append the absent
// ones to the accumulator's units -- slab by slab when the plan is slab-wise: one scan of the reads, then for every
// slab of level-1 bins the second multisplit level and the membership kernel over the slab's segments.  No host
// synchronisation before the end: a kernel that loses records (the partition's overflow list is full) raises
// Counters::n_overflow / fail_mark on the device and every later membership launch of the batch returns at once
// (batch_has_failed), so what reached the accumulator is exactly the slabs before the failing one.
// Returns DK_OK, or DK_ERR_OVERFLOW with
//   *fatal = true:  the accumulator itself lost records (units and overflow list full): it is unusable
//   *fatal = false: the partition lost records in slab F; every k-mer of the window with hash < *redo_from has been
//                   appended (*absent_done of them) and the caller redoes hashes >= *redo_from through the direct family
template <bool WIDE>
inline dk_status bucketed_accum_add_t(dk_engine *e, dk_accum *a, const dk_reads *r, bool *fatal, uint64_t *redo_from,
                                      uint64_t *absent_done)
{
    using R = typename RecOf<WIDE>::type;
    *fatal = false;
    *redo_from = a->wbits ? (uint64_t)a->widx << (64 - a->wbits) : 0;
    *absent_done = 0;
    BucketPlan p;
    const int sbits = a->s ? pick_sub_bits(e, a->T - a->wbits) : 0;
    if (!make_plan(e, r, &p, a->T, a->wbits, sbits, true)) return fail(e, DK_ERR_UNSUPPORTED, "no bucketed plan for this geometry");
    BucketBufs<R> B;
    dk_status st = bucketed_partition<WIDE>(e, r, p, B, a->wbits, a->widx, false);
    if (st != DK_OK) { free_bufs(e, B); return st; }
    const MissOut<R> mo = accum_out<R>(e, a);
    const uint64_t win_seg0 = (uint64_t)a->widx << (a->T - a->wbits);
    const uint64_t regions_per_slab = p.n_seg / p.slabs, segs_per_slab = regions_per_slab << p.sbits;
    const char *probe_name = a->s && a->s->exact ? "seg_exact_probe" : a->s ? "seg_probe" : "seg_append";
    hipError_t h = hipSuccess;
    for (uint32_t sl = 0; sl < p.slabs && h == hipSuccess; sl++) {
        if (p.slabs > 1) {
            launch_repart<WIDE>(e, p, B, a->wbits, sl);
            stage_mark(e, "repart");
        }
        PieceList<R> list = slab_piece_list(p, B, sl, a->wbits, a->widx);
        list.sbits = p.sbits;
        list.sub_shift = 64 - a->T;
        // the slab's first unit inside the window: the sink addresses units from there
        MissOut<R> ms = mo;
        const uint64_t unit0 = ((uint64_t)sl * segs_per_slab) << a->u;
        ms.cnt = mo.cnt + unit0;
        ms.recs = (R *)((char *)mo.recs + unit0 * (uint64_t)a->unit_cap * accum_rec_bytes(a));
        const uint64_t seg_base = win_seg0 + (uint64_t)sl * segs_per_slab;
        bool launched = false;
        if constexpr (!WIDE) {
            launched = true;
            if (a->packed && p.packed2) h = launch_seg_probe<R, ACC_PACKED, true>(e, a->s, list, segs_per_slab, a->T, seg_base, ms);
            else if (a->packed) h = launch_seg_probe<R, ACC_PACKED, false>(e, a->s, list, segs_per_slab, a->T, seg_base, ms);
            else if (p.packed2) h = launch_seg_probe<R, ACC_PLAIN, true>(e, a->s, list, segs_per_slab, a->T, seg_base, ms);
            else launched = false;
        }
        if (!launched) h = launch_seg_probe<R, ACC_PLAIN>(e, a->s, list, segs_per_slab, a->T, seg_base, ms);
        if (h == hipSuccess) stage_mark(e, probe_name);
    }
    if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "seg_probe launch failed: %s", hipGetErrorString(h));
    if (st == DK_OK) st = sync_counters(e, "bucketed accumulate");
    uint64_t h_lim = 0;                         // != 0: the partition failed in a later slab; hashes below are complete
    if (st == DK_ERR_OVERFLOW) {
        if (e->h_ctr->n_sink_drop) { *fatal = true; free_bufs(e, B); return st; }
        const uint64_t F = 0xFFFFFFFFULL - (e->h_ctr->fail_mark & 0xFFFFFFFFULL);
        const uint64_t bins_done = e->h_ctr->fail_mark && F < p.slabs ? F * p.slab_bins : 0;
        if (!bins_done) { free_bufs(e, B); return st; }     // nothing is complete: the whole window is redone
        h_lim = (((uint64_t)a->widx << p.b1) | bins_done) << (64 - a->wbits - p.b1);
        *redo_from = h_lim;
        st = DK_OK;
        stage_mark(e, "slab_partial");
    }
    // overflow records of the partition (normally none): probe one by one, append the absent ones through global cursors
    if (st == DK_OK && e->h_ctr->n_ovf) {
        const uint64_t n_ovf = std::min<uint64_t>(e->h_ctr->n_ovf, B.ovf_cap);
        st = pool_alloc(e, n_ovf * sizeof(R), (void **)&B.ovf_miss);
        if (st == DK_OK) {
            const OvfList<R> ovf{B.ovf, &e->d_ctr->n_ovf, B.ovf_cap};
            ovf_probe_kernel<R><<<grid_for(e, n_ovf, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                a->s ? a->s->d_words : nullptr, ovf, (int)e->cfg.filter_log2_bits - 9, (int)e->cfg.n_hashes,
                a->s && a->s->exact ? a->T : 0, 1, 0, B.ovf_miss, nullptr, e->d_ctr, h_lim);
            h = hipGetLastError();
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow probe failed: %s", hipGetErrorString(h));
        }
        if (st == DK_OK) {
            // (the partition's n_overflow is still raised after a partial failure: read the counters without judging them)
            h = hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream);
            if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow probe failed: %s", hipGetErrorString(h));
        }
        if (st == DK_OK && e->h_ctr->n_ovf_miss) {
            const uint64_t n_om = e->h_ctr->n_ovf_miss;
            if (a->packed && !WIDE)
                acc_append_kernel<R, !WIDE><<<grid_for(e, n_om, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                    B.ovf_miss, n_om, accum_unit_bits(a), accum_unit_base(a), mo, e->d_ctr);
            else
                acc_append_kernel<R, false><<<grid_for(e, n_om, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                    B.ovf_miss, n_om, accum_unit_bits(a), accum_unit_base(a), mo, e->d_ctr);
            h = hipGetLastError();
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow append failed: %s", hipGetErrorString(h));
            else stage_mark(e, "ovf_append");
        }
        if (st == DK_OK) {
            h = hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream);
            if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow append failed: %s", hipGetErrorString(h));
            else if (e->h_ctr->n_sink_drop) { *fatal = true; st = fail(e, DK_ERR_OVERFLOW, "accumulator full"); }
        }
        if (st == DK_OK) e->h_ctr->n_absent += e->h_ctr->n_ovf_miss;
    }
    free_bufs(e, B);
    if (st == DK_OK && h_lim) {
        *absent_done = e->h_ctr->n_absent;
        return DK_ERR_OVERFLOW;
    }
    if (st != DK_OK && st != DK_ERR_UNSUPPORTED) *fatal = true;     // died half-way: appended or not is unknown
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
