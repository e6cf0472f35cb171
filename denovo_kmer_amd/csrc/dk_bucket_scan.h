// dk_bucket_scan.h -- scan_part (extraction + hash + level-1 multisplit) and kmers_tile (extraction only)
// (part of the bucketed kernel family: dk_kernels_bucket.h has the overview and includes the parts in order)
#pragma once
#include "dk_bucket_common.h"

namespace dk {

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
// Window-major thread mapping (k <= 32, batches whose reads all have one length L: a flag at every position that is L modulo
// L + 1, verified on the device when the batch was made -- `ok`).  A read of L bases has L - k + 1 windows but L + 1 stream
// positions: position by position, 16 per thread, 21 % of the slots of a 150-bp read at k = 31 hash a window that cannot
// exist (the last k - 1 positions of a read and its separator).  Here a read is dealt to `tpr` threads of `wpt` <= 16
// consecutive windows each (150 bp, k = 31: 8 threads x 15 windows), so every slot but the rounding holds a real window.
// The parts cover the offsets 0 .. min(tpr * wpt, L + 1) of their read; whatever starts at an offset in [L - k + 1, L] covers
// the flagged position L, so no valid window is missed and none is taken twice -- whatever the flags at the positions
// = L mod L + 1 stand for (a separator, or an N).
struct WindowMajor {
    uint32_t stride;               // L + 1; 0 = position-major (the general mapping)
    uint32_t tpr, wpt;
    uint32_t n_slots;              // n_reads * tpr (< 2^31)
    const uint32_t *ok;            // device flag: 1 = every position = L mod stride is flagged
};

template <int THREADS, int PER_THREAD, int MIN_WAVES, bool WIDE, bool WINDOWED>
__global__ void __launch_bounds__(THREADS, MIN_WAVES)
scan_part_kernel(StreamView s, int k, int canonical, uint64_t seed, int b1, uint32_t capw,
                 typename RecOf<WIDE>::type *__restrict__ out, uint32_t *__restrict__ cnt1, uint32_t n_tiles,
                 OvfList<typename RecOf<WIDE>::type> ovf, Counters *ctr, int wbits, uint32_t widx, uint64_t bin_stride,
                 uint64_t piece_stride, WindowMajor wmv)
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
    uint32_t n_records = 0;
    uint32_t n_all = 0;                                   // WINDOWED: valid windows inside or outside the window
    multisplit_init(L, nbins);
    Stamps st;
#ifdef DK_DEBUG_INFO
    if (blockIdx.x == 0 && threadIdx.x == 0) {           // where the kernel's code was loaded (instruction-cache alignment studies)
        unsigned long long pc;
        asm volatile("s_getpc_b64 %0" : "=s"(pc));
        ctr->dbg[7] = pc;
    }
#endif

    const uint64_t last_b = s.n_bwords - 1, last_m = s.n_mwords - 1;
    // window-major (k <= 32 only; decided once per launch from the batch's device flag): tiles count thread slots, not positions
    const bool wm = !WIDE && wmv.stride != 0 && *wmv.ok != 0;
    if (wm) n_tiles = (wmv.n_slots + THREADS - 1) / THREADS;
    const uint32_t n_win = wm ? wmv.wpt : (uint32_t)PER_THREAD;          // windows a thread hashes per tile
    // first position of the thread's stretch in `tile`; *lim (if given) = how many of its windows are the thread's own
    auto pos_of = [&](uint32_t tile, uint32_t *lim = nullptr) -> uint64_t {
        if (wm) {
            const uint32_t g = tile * (uint32_t)THREADS + (uint32_t)tid;
            if (g >= wmv.n_slots) return s.n_bases;                       // beyond the last read: nothing valid
            const uint32_t r = g / wmv.tpr, off = (g - r * wmv.tpr) * wmv.wpt;
            // the read's last part stops at the read's own positions: what starts beyond belongs to the next read's first part
            if (lim) *lim = wmv.stride - off < wmv.wpt ? wmv.stride - off : wmv.wpt;
            return (uint64_t)r * wmv.stride + off;
        }
        return (uint64_t)tile * TILE + (uint64_t)tid * PER_THREAD;
    };
    // (k <= 32: the third bases word is needed only where a thread's stretch starts deep inside a word, i.e. window-major;
    // its top half is enough -- 16 windows + k - 1 bases end at most 93 bits after the stretch's first bit)
    auto load_words = [&](uint32_t tile, uint64_t &w0, uint64_t &w1, uint64_t &w2, uint64_t &m0, uint64_t &m1) {
        const uint64_t p0 = pos_of(tile);
        const uint64_t bw = p0 >> 5, mw = p0 >> 6;
        w0 = s.bases[bw < last_b ? bw : last_b];
        w1 = s.bases[bw + 1 < last_b ? bw + 1 : last_b];
        if constexpr (WIDE) w2 = s.bases[bw + 2 < last_b ? bw + 2 : last_b];
        else w2 = wm ? (uint64_t)((const uint32_t *)s.bases)[2 * (bw + 2 < last_b ? bw + 2 : last_b) + 1] : 0;   // the word's top half
        m0 = s.mask[mw < last_m ? mw : last_m];
        m1 = s.mask[mw + 1 < last_m ? mw + 1 : last_m];
    };
    const int sk = (WIDE ? 128 : 64) - 2 * k;            // right-alignment shift of a window
    const uint64_t kmask_shift = 64 - k;
    const uint64_t G = gridDim.x, w = blockIdx.x;
    // Piece w of bin b starts at record b * bin_stride + w * piece_stride of `out` (Level1Layout, dk_bucket_host.h):
    // bin-major (a bin's pieces side by side) or workgroup-major (a workgroup's pieces side by side: its 2^b1 write
    // frontiers then lie within 2^b1 * capw records instead of being spread over the whole buffer -- far fewer pages)
    const uint64_t piece_base = w * piece_stride;

    // the tile being hashed: stream left-aligned at p0 -- bases in (v0, v1[, v2]), flags in (mh, ml)
    uint64_t p0 = 0, v0 = 0, v1 = 0, v2 = 0, mh = 0, ml = 0, rch = 0, rcl = 0;
    uint32_t okbits = 0;                                   // k <= 32: bit (PER_THREAD - 1 - j) = window j is a k-mer
    auto prep = [&](uint32_t tile, uint64_t w0, uint64_t w1, uint64_t w2, uint64_t m0, uint64_t m1) {
        uint32_t lim = n_win;
        p0 = pos_of(tile, &lim);
        const int o = 2 * (int)(p0 & 31);                 // position-major: PER_THREAD 8: 0,16,32,48; 16: 0,32; window-major: any
        v0 = o ? (w0 << o) | (w1 >> (64 - o)) : w0;
        // (k <= 32: w2 holds the TOP half of the third word in its low 32 bits)
        v1 = o ? (w1 << o) | ((WIDE ? w2 : w2 << 32) >> (64 - o)) : w1;
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
            if (wm) okbits &= ~((1u << (PER_THREAD - lim)) - 1u);          // the thread's part of its read: lim <= n_win windows
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
        // which of the thread's windows became records: the plain k <= 32 scan has that in okbits (bit PER_THREAD - 1 - j) until
        // the next tile is prepared, which is after the scatter; the other shapes decide per window and keep their own mask
        constexpr bool KEEP_VALID = WIDE || WINDOWED;
        uint32_t valid = 0;
        // count phase of the first tile
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            uint32_t r = 0;
            if ((uint32_t)j < n_win && window(j, hs[j])) {
                if constexpr (KEEP_VALID) valid |= 1u << j;
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
                for (int v = 0; v < L.N_BINS / 64 - 1; v++) {
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
                if (KEEP_VALID ? (valid >> j) & 1u : (okbits >> (PER_THREAD - 1 - j)) & 1u) L.stage[L.off[bin_of(hs[j].h)] + ((rk[j / 2] >> (16 * (j & 1))) & 0xffffu)] = hs[j];
            lds_barrier();                               // C: stage ready, cnt[] zero
            st.mark(2);
            const uint32_t total = L.total;
            const uint32_t total_s = __builtin_amdgcn_readfirstlane(total);
            const bool checked = L.ovf_seen != 0;        // some piece may be full: bounds check + overflow list
            const bool has_next = tile + gridDim.x < n_tiles;
            if (has_next) {
                const uint64_t w0 = nw0, w1 = nw1, w2 = nw2, m0 = nm0, m1 = nm1;
                if (tile + 2 * gridDim.x < n_tiles) load_words(tile + 2 * gridDim.x, nw0, nw1, nw2, nm0, nm1);
                prep(tile + gridDim.x, w0, w1, w2, m0, m1);
            }
            valid = 0;
            // copy-out of this tile, interleaved with the count phase of the next one.  Per record three LDS round trips (the
            // staged record, the rank of the next tile's window, the bin's write pointer): the staged record is requested first
            // and needed only after the window has been hashed.  The rank is packed outside the `checked` branch: both arms
            // writing rk[] cost a copy of the whole array at their join, five moves per record.
#pragma unroll
            for (int j = 0; j < PER_THREAD; j++) {
                const uint32_t i = (uint32_t)j * THREADS + tid;
                const bool mine = i < total;
                const R rec = L.stage[i];                  // (i < TILE: beyond `total` a stale record, never stored)
                __builtin_amdgcn_sched_barrier(0);
                uint32_t r;                                // the rank of a window that is no k-mer is never looked at: any value
                asm volatile("" : "=v"(r));
                if (has_next) {
                    if ((uint32_t)j < n_win && window(j, hs[j])) {
                        if constexpr (KEEP_VALID) valid |= 1u << j;
                        r = atomicAdd(&L.cnt[bin_of(hs[j].h)], 1u);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                const uint32_t bin = bin_of(rec.h);
                rk[j / 2] = (j & 1) ? __builtin_amdgcn_perm(r, rk[j / 2], 0x05040100u) : r;     // (low halves of both: one v_perm_b32)
                __builtin_amdgcn_sched_barrier(0);
                if (!checked) {
                    // (all but one of a tile's 16 strides lie wholly below `total`: decided on the scalar unit)
                    if ((uint32_t)(j + 1) * THREADS <= total_s) store_global((R *)(uintptr_t)L.gptr[bin] + i, rec);
                    else if (mine) store_global((R *)(uintptr_t)L.gptr[bin] + i, rec);
                } else {
                    const uint32_t idx = i + L.delta[bin];   // 32-bit on purpose: delta is a wrapped difference
                    if (mine && idx < capw) out[(uint64_t)bin * bin_stride + piece_base + idx] = rec;
                    uint32_t lost = 0;                       // (rare arm: tallied in LDS, so that no register crosses the arms' join)
                    ovf_append(ovf, mine && idx >= capw, rec, lost);
                    if (lost) atomicAdd(&L.ovf_lost, lost);
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
    if (tid == 0 && L.ovf_lost) {                             // (multisplit_finish begins with a workgroup barrier)
        atomicAdd(&ctr->n_overflow, (unsigned long long)L.ovf_lost);
        atomicMax(&ctr->fail_mark, 0xFFFFFFFFULL);           // records lost before any slab: everything is to be redone
    }
}

// ---- kmer.rs stand-in at streaming speed: canonical k-mer / hash / not-a-k-mer bit per position ------------
// Same window machinery as scan_part (a thread owns 16 consecutive positions, two or three register
// words cover all its windows, the reverse complement rolls), but nothing is partitioned: every wave
// transposes its 1024 results through its own 8.5 KiB of LDS so that each store instruction writes 64
// consecutive positions.  No workgroup barrier anywhere.
template <int THREADS, bool WIDE, bool NT = false>
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
                if (p < s.n_bases) {
                    if constexpr (NT) __builtin_nontemporal_store(v, &dst[p]);       // (streamed out, never read back here)
                    else dst[p] = v;
                }
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

}  // namespace dk
