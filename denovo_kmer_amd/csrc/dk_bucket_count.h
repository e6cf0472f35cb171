// dk_bucket_count.h -- seg_count: exact counting of the absent records of one counting unit
// (part of the bucketed kernel family: dk_kernels_bucket.h has the overview and includes the parts in order)
#pragma once
#include "dk_bucket_seg.h"

namespace dk {

// Exact counting of one segment's absent records.
// Most absent k-mers are singletons (sequencing errors), so a hash table for all of them is wasted
// work.  Two 64-Kbit LDS bitmaps classify the records first: bit(h) set twice => the record MAY have
// a twin (true duplicate or bitmap collision) and goes through a small LDS hash table; every other
// record is provably unique and is emitted with count 1 straight from registers.  Exact for any
// input: all copies of a k-mer share a bit, so all of them are flagged.
// Two geometries: <512 threads, 2048 slots, 64-Kbit bitmaps> for segments with thousands of absent
// records, <128, 512, 8 Kbit> (8 KB of LDS, many workgroups per CU) when a segment holds a few hundred.
// PACKED (k <= 32 only): the pieces are units of an accumulator's packed store (6 bytes per record, dk_bucket_seg.h); the
// extra list (the accumulator's overflow records) holds plain records.
template <int CNT_THREADS, int CNT_SLOTS, int CNT_BM_WORDS, bool WIDE, bool PACKED = false, int FAST = 0>
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
    __shared__ uint32_t fcount;                          // few-flagged path: records gathered so far
    __shared__ unsigned long long gbase;
    constexpr bool TWO_POS = WIDE;                       // two bitmap positions per record (see bit2_of): pays where the table phase is dear
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
    // Which round of the table phase a key belongs to: 20 hash bits ABOVE the table's slot bits (8..18) and BELOW the unit's
    // common prefix -- k <= 32 keys are the hashes themselves, and all hashes of a unit share their top T bits.  (Taken from
    // bits 36..55, as until round 3, the selector was constant within a unit for T > 8 + 20: every split of a crowded round
    // put all keys into one sub-round again, the rounds multiplied to 4096 and more, each a pass over the unit -- a unit whose
    // records are mostly duplicates, e.g. a sample seen twice, took 6 ms instead of 20 us: 3 s against 12 ms for 2 x 10^9 records.)
    auto round_of = [](unsigned long long f, uint32_t n_rounds) -> uint32_t {
        return (uint32_t)((((f >> 19) & 0xFFFFF) * (uint64_t)n_rounds) >> 20);
    };
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
    Stamps st;
    for (uint64_t seg_id = blockIdx.x; seg_id < n_seg; seg_id += gridDim.x) {
        const SegPieces<R> sp = seg_pieces(pl, seg_id);
        // FAST (PACKED, one piece per unit; chosen by the host): 1 = no extra records anywhere, the unit is its packed blocks;
        // 2 = the unit's index space is its packed blocks, whole, followed by its extra records -- [0, n_packed) records,
        // [n_packed, np64) padding of the last block, [np64, n) extras
        const uint32_t n_packed = sp.start[MAX_R], np64 = FAST == 2 ? (n_packed + 63u) & ~63u : n_packed;
        const uint32_t n = FAST == 2 ? np64 + sp.n_extra : sp.total();
        if (n == 0) continue;
        const unsigned long long EMPTY = WIDE ? 0ULL : (unsigned long long)((seg_id + unit_base) ^ 1ULL) << (64 - T);
        const uint32_t n_chunks = (n + CNT_CHUNK - 1) / CNT_CHUNK;
        const bool single = n_chunks == 1;                 // the common case: the records stay in registers
        R hv[CNT_RPT];
        const uint64_t prefix = (seg_id + unit_base) << (64 - T);      // PACKED: the hash bits every record of the unit shares
        auto rec_at = [&](uint32_t i) -> R {
            if constexpr (PACKED && !WIDE) {
                if (i >= sp.start[MAX_R]) return sp.extra[i - sp.start[MAX_R]];
                uint32_t r = 0, st = 0;
                if (!sp.single) {
#pragma unroll
                    for (int q = 1; q < MAX_R; q++)
                        if (i >= sp.start[q]) { r = (uint32_t)q; st = sp.start[q]; }
                }
                const uint64_t unit = pl.n_segs ? (uint64_t)r * pl.n_segs + seg_id : seg_id * pl.n_pieces + r;
                return R{packed_load(pl.recs, unit, pl.piece_cap, i - st, prefix)};
            } else {
                return sp.at(i);
            }
        };
        // PACKED, one piece and no overflow records (an accumulator counted on its own GPU): record i of the unit sits in block
        // i / 64 at lane i % 64, and a wave's 64 records of one register slot are exactly one block -- the block's address is
        // wave-uniform, the lane's offset inside it the same for every slot: no per-record address arithmetic at all
        // (FAST: chosen by the host when the list has one piece per unit and no extra records)
        auto load_chunk = [&](uint32_t c) {
            if constexpr (PACKED && !WIDE && FAST) {
                {
                    const uint64_t unit = pl.n_segs ? seg_id : seg_id * pl.n_pieces;
                    const char *ub = (const char *)pl.recs + unit * (uint64_t)pl.piece_cap * PACKED_REC_BYTES;
                    const uint32_t wave_s = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6), lane = (uint32_t)tid & 63u;
                    const uint32_t o32 = lane * 4u, o16 = (uint32_t)PACKED_BLOCK_RECS * 4u + lane * 2u;
#pragma unroll
                    for (int u = 0; u < CNT_RPT; u++) {
                        const uint32_t blk = (c * CNT_CHUNK + (uint32_t)u * CNT_THREADS) / 64u + wave_s;      // wave-uniform
                        if (blk * 64u >= n) continue;
                        if (FAST == 1 || blk * 64u < np64) {   // (a block the unit has begun is allocated whole: its last lanes read padding)
                            const char *b = ub + (uint64_t)blk * PACKED_BLOCK_BYTES;
                            hv[u].h = prefix | ((uint64_t)*(const uint16_t *)(b + o16) << 32) | *(const uint32_t *)(b + o32);
                        } else {                               // the unit's share of the overflow list (rare)
                            const uint32_t x = blk * 64u + lane - np64;
                            hv[u] = sp.extra[x < sp.n_extra ? x : 0];
                        }
                    }
                    return;
                }
            }
#pragma unroll
            for (int u = 0; u < CNT_RPT; u++) {
                const uint32_t i = c * CNT_CHUNK + (uint32_t)u * CNT_THREADS + tid;
                hv[u] = rec_at(i < n ? i : 0);
            }
        };
        auto have_at = [&](uint32_t c, int u) -> bool {
            const uint32_t i = c * CNT_CHUNK + (uint32_t)u * CNT_THREADS + tid;
            if constexpr (FAST == 2) return i < n_packed || (i >= np64 && i < n);
            else return i < n;
        };
        // FAST, single chunk: which of the thread's slots hold a record, worked out once (the test runs in every pass)
        uint32_t hmask = 0;
        if constexpr (FAST == 2) {
            if (single) {
#pragma unroll
                for (int u = 0; u < CNT_RPT; u++) hmask |= (have_at(0, u) ? 1u : 0u) << u;
            }
        }
        auto have = [&](uint32_t c, int u) -> bool {
            if constexpr (FAST == 2) {
                if (single) return (hmask >> u) & 1u;
            }
            return have_at(c, u);
        };
        // bitmap of >= 16 bits per record where the geometry has them (<= 6 % of the unique records collide and take the
        // table path), a power of two up to CNT_BM_WORDS
        uint32_t bm_words = 64;
        while (bm_words * 2 < n && bm_words < (uint32_t)CNT_BM_WORDS) bm_words <<= 1;
        const uint32_t bm_mask = bm_words * 32 - 1;
        auto bit_of = [=](const R &rec, uint32_t &w, uint32_t &m) {
            // (the record's hash, not its table fingerprint: equal k-mers have equal hashes, which is all the bitmaps need, and
            // the k > 32 fingerprint costs an fmix64 -- now paid by the flagged records only)
            const uint32_t b = (uint32_t)(rec.h >> 20) & bm_mask;
            w = b >> 5;
            m = 1u << (b & 31);
        };
        // A second position from other hash bits: a record goes to the table only if BOTH of its positions were hit twice.
        // All copies of a k-mer share both positions, so duplicates are still flagged, every one of them; a unique record is
        // flagged only when two other records hit its two positions -- (2 n / bits)^2 instead of n / bits: 0.5 % instead of
        // 3.4 % at 30 bits per record.  k > 32 only, where the table phase (fingerprints, a re-check pass: 58 % of the kernel's
        // time at configs[4]) is what the flagged records cost: 19.5 -> 17.8 ms there; with 8-byte records the second
        // position costs more than the smaller table saves (4.34 -> 4.48 ms per whole-genome step).
        auto bit2_of = [=](const R &rec, uint32_t &w, uint32_t &m) {
            const uint32_t b = (uint32_t)(rec.h >> 2) & bm_mask;
            w = b >> 5;
            m = 1u << (b & 31);
        };
        // with the records in registers (single) the verdict of pass 2 is kept (fbits): passes 3 and 4 would otherwise re-derive it
        constexpr bool KEEP_FLAGS = !(WIDE && CNT_THREADS == 1024);     // (that geometry has no register to spare)
        uint32_t fbits = 0;                                // bit u: record u of this thread may have a twin (single only)
        for (uint32_t i = tid; i < bm_words; i += CNT_THREADS) { bm_a[i] = 0; bm_b[i] = 0; }
        if (tid == 0) fcount = 0;
        if (single) load_chunk(0);
        lds_barrier();
        st.mark(0);                                   // (DK_STAMPS) bitmaps cleared, records in registers
        // pass 1: mark
        for (uint32_t c = 0; c < n_chunks; c++) {
            if (!single) load_chunk(c);
#pragma unroll
            for (int u = 0; u < CNT_RPT; u++) {
                if (!have(c, u)) continue;
                uint32_t w, m;
                bit_of(hv[u], w, m);
                if (atomicOr(&bm_a[w], m) & m) atomicOr(&bm_b[w], m);
                if constexpr (TWO_POS) {
                    bit2_of(hv[u], w, m);
                    if (atomicOr(&bm_a[w], m) & m) atomicOr(&bm_b[w], m);
                }
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
                bit_of(hv[u], w, m);
                bool fl = (bm_b[w] & m) != 0;
                if constexpr (TWO_POS) {
                    bit2_of(hv[u], w, m);
                    fl = fl && (bm_b[w] & m) != 0;
                }
                if (fl) my_flagged++; else my_unique++;
                if constexpr (KEEP_FLAGS) fbits |= (fl ? 1u : 0u) << u;     // (used for a single chunk only)
            }
        }
        auto flagged = [&](int u, const R &rec) -> bool {
            if constexpr (KEEP_FLAGS) {
                if (single) return (fbits >> u) & 1u;   // the verdict of pass 2, kept: one bit per record instead of a hash + an LDS read
            }
            uint32_t w, m;
            bit_of(rec, w, m);
            bool fl = (bm_b[w] & m) != 0;
            if constexpr (TWO_POS) {
                bit2_of(rec, w, m);
                fl = fl && (bm_b[w] & m) != 0;
            }
            return fl;
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
        st.mark(1);                                   // marked, classified, bases known, output reserved
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
        st.mark(2);                                   // unique records emitted
        // Flagged records: exact counts in the LDS hash table, one sub-range of the key space per
        // round.  The number of rounds starts from a guess (8 copies per key) and a round whose keys
        // do not fit is split in four and redone -- nothing of it has been emitted yet -- so a segment
        // holding a million copies of one k-mer costs one round, not a thousand.
        // Few flagged records (at most 64, all in registers): no table.  Every wave appends its flagged records to one LDS list;
        // wave 0 then compares each with all of them -- the full key, so nothing needs a re-check -- counts its copies, and the
        // first copy of every k-mer reports it.  One barrier instead of the table's six, no clearing, no second walk; the
        // other waves go on to the next unit (the list is not written again before wave 0 has passed two more barriers).
        constexpr bool FEW = GATHER;
        bool few_done = false;
        if (FEW && n_flagged && n_flagged <= 64 && single) {
            R *const flist = &wbuf[0][0];                  // (CNT_THREADS / 64 * WB >= 64 entries)
#pragma unroll
            for (int u = 0; u < CNT_RPT; u++) {
                const bool want = have(0, u) && flagged(u, hv[u]);
                if (__ballot(want) == 0) continue;
                if (want) flist[atomicAdd(&fcount, 1u)] = hv[u];
            }
            lds_barrier();
            if (wave == 0) {
                const uint32_t lane = (uint32_t)lane_id(), nf = n_flagged;
                const R mine = flist[lane < nf ? lane : 0];
                uint32_t copies = 0;
                bool first = lane < nf;
                for (uint32_t j = 0; j < nf; j++) {
                    const R o = flist[j];                  // (one address for all lanes: a broadcast read)
                    bool eq = o.h == mine.h;
                    if constexpr (WIDE) eq = eq && rec_hi(o) == rec_hi(mine);
                    copies += eq ? 1u : 0u;
                    if (eq && j < lane) first = false;
                }
                const bool keep = first && copies >= min_count;
                const uint64_t kb = __ballot(keep);
                const uint32_t n_first = (uint32_t)__popcll(__ballot(first));     // (by every lane: not inside the conditional below)
                n_distinct += lane == 0 ? n_first : 0u;
                if (kb) {
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(fill, (unsigned long long)__popcll(kb));
                    base = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)(base >> 32)) << 32) |
                           (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
                    if (keep) {
                        const uint64_t o = base + (uint64_t)popc_below(kb);
                        if (o < region_cap) {
                            out_kmer[region_base + o] = rec_lo(mine, seed);
                            if constexpr (WIDE) out_hi[region_base + o] = rec_hi(mine);
                            out_cnt[region_base + o] = copies;
                        } else {
                            n_fail++;
                        }
                    }
                }
            }
            few_done = true;
        }
        if (n_flagged && !few_done) {
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
                                want = round_of(fu, rounds) == r;
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
                            const uint32_t rr = round_of(f, rounds);
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
        if (!few_done) lds_barrier();                 // (the table and its counters are cleared by the next unit)
        st.mark(3);                                   // flagged records counted and emitted
    }
    st.flush(ctr, 4);
    n_distinct = (uint32_t)wave_sum(n_distinct);
    n_fail = (uint32_t)wave_sum(n_fail);
    if (lane_id() == 0) {
        if (n_distinct) atomicAdd(&ctr->n_distinct, (unsigned long long)n_distinct);
        if (n_fail) atomicAdd(&ctr->n_overflow, (unsigned long long)n_fail);
    }
}

}  // namespace dk
