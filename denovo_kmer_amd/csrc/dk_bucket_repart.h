// dk_bucket_repart.h -- repart: the second (and third) multisplit level
// (part of the bucketed kernel family: dk_kernels_bucket.h has the overview and includes the parts in order)
#pragma once
#include "dk_bucket_common.h"

namespace dk {

// ---- level 2: one workgroup per tile of a level-1 piece; records go to per-SEGMENT regions through
// global cursors.  Shared write frontiers keep the DRAM pages and L2 lines being written few and
// hot (every resident workgroup appends to the same 2^b2 segments of one coarse bin at a time),
// which measured faster than private level-2 pieces; the cursor atomics are issued before the
// scatter phase and only waited for after it, so their latency is covered.
// PK (k <= 32, regions of >= 16 prefix bits): the regions receive packed 6-byte records (dk_bucket_common.h)
// CONCAT (level 2 over the scan's pieces, G <= 512): the G pieces of a bin are read as ONE array -- tile t of the bin is
// records [t * TILE, (t + 1) * TILE) of their concatenation, found through a prefix sum of the piece sizes -- so every tile
// but a bin's last is full; `tiles_per_piece` then is the number of tiles launched per BIN (enough for G full pieces: a
// bin cannot hold more; the workgroups beyond the bin's records return after the prefix sum).  Piece by piece, a 22 K-record
// piece filled its three 8 K tiles to 90 % and launched a fourth that returned at once (whole-genome child step: 1024 bins x
// 256 pieces).
template <int THREADS, int PER_THREAD, int MIN_WAVES, class R, bool PK = false, bool CONCAT = false>
__global__ void __launch_bounds__(THREADS, MIN_WAVES)
repart_kernel(const R *__restrict__ in, const uint32_t *__restrict__ cnt1, uint32_t G, uint32_t capw,
              uint32_t tiles_per_piece, int b1, int b2, uint32_t cap2, R *__restrict__ out,
              uint32_t *__restrict__ cursor2, OvfList<R> ovf, Counters *ctr, int xcd_affine = 0, uint64_t bin_stride = 0,
              uint64_t piece_stride = 0, uint32_t bin0 = 0, uint32_t slab = 0)
{
    // bin0 / slab: slab-wise partition (bucketed_partition) -- this launch covers the level-1 bins from bin0 on, and `out`
    // holds the regions of these bins only (region (bin0 << b2) first); cursor2 is indexed by the region's global number
    constexpr int TILE = THREADS * PER_THREAD;
    constexpr int NB = THREADS >= MAX_BINS2 ? MAX_BINS2 : MAX_BINS;      // the bin scan is one thread per bin
    __shared__ SplitLds<THREADS, PER_THREAD, R, NB, false> L;
    const int tid = (int)threadIdx.x;
    // one-dimensional grid (the number of bins can exceed the 65535 of grid.y): bin-major, then piece, then tile
    const uint32_t per_bin = CONCAT ? tiles_per_piece : G * tiles_per_piece;
    uint32_t b = blockIdx.x / per_bin, bx = blockIdx.x % per_bin;
    if (xcd_affine) {
        // eight bins at a time, one per XCD (consecutive blocks are dealt round-robin over the XCDs): all tiles of a bin
        // then append to its 2^b2 frontiers through ONE L2, which assembles whole lines (speed only)
        const uint32_t slot = blockIdx.x >> 3;
        b = 8 * (slot / per_bin) + (blockIdx.x & 7);
        bx = slot % per_bin;
    }
    b += bin0;
    __shared__ uint32_t pstart[CONCAT ? 513 : 1];        // CONCAT: first record of every piece in the bin's concatenation
    __shared__ uint32_t pscan[CONCAT ? 17 : 1];
    uint32_t n, t0;
    const R *src = nullptr;
    if constexpr (CONCAT) {
        uint32_t c = 0;
        if ((uint32_t)tid < G) {
            c = cnt1[(uint64_t)b * G + tid];
            if (c > capw) c = capw;
        }
        const uint32_t ex = block_excl_scan(c, pscan, &pscan[16]);
        if ((uint32_t)tid < G) pstart[tid] = ex;
        if (tid == 0) pstart[G] = pscan[16];
        __syncthreads();
        n = pstart[G];
        t0 = bx * TILE;
    } else {
        const uint32_t w = bx / tiles_per_piece;
        t0 = (bx % tiles_per_piece) * TILE;
        const uint64_t piece = (uint64_t)b * G + w;
        n = cnt1[piece];
        if (n > capw) n = capw;
        src = bin_stride ? in + (uint64_t)b * bin_stride + (uint64_t)w * piece_stride : in + piece * capw;     // (level 3: plain pieces)
    }
    if (t0 >= n) return;
    const int nbins = 1 << b2;
    const int shift = 64 - b1 - b2;
    auto bin_of = [=](uint64_t h) -> uint32_t { return (uint32_t)(h >> shift) & (uint32_t)(nbins - 1); };
    for (int i = tid; i < NB; i += THREADS) L.cnt[i] = 0;
    if (tid == 0) L.ovf_seen = 0;
    R hs[PER_THREAD];
    if constexpr (CONCAT) {
        // piece of the thread's first record by bisection; the later ones lie THREADS records further each: at most one
        // step on per record in the usual case (pieces of thousands of records), bisection again otherwise.  Positions
        // beyond the bin's end read its last record (ignored below), which keeps the walk monotonic.
        const R *bin_base = in + (uint64_t)b * bin_stride;
        auto bisect = [&](uint32_t r) -> uint32_t {
            uint32_t lo = 0, hi = G;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (pstart[mid] <= r) lo = mid; else hi = mid;
            }
            return lo;
        };
        uint32_t w = 0;
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            const uint32_t i = t0 + (uint32_t)j * THREADS + tid;
            const uint32_t r = i < n ? i : n - 1;
            if (j == 0) {
                w = bisect(r);
            } else if (pstart[w + 1] <= r) {               // (pstart[G] = n > r: w + 1 <= G here)
                w++;
                if (pstart[w + 1] <= r) w = bisect(r);
            }
            hs[j] = bin_base[(uint64_t)w * piece_stride + (r - pstart[w])];
        }
    } else {
#pragma unroll
        for (int j = 0; j < PER_THREAD; j++) {
            const uint32_t i = t0 + (uint32_t)j * THREADS + tid;
            hs[j] = src[i < n ? i : 0];
        }
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
    const uint64_t seg0 = (uint64_t)(b - bin0) << b2;
    uint32_t n_overflow = 0;
    if (!L.ovf_seen) {
        // every segment region still has room for this tile: no bounds check, no overflow ballot
#pragma unroll 4
        for (uint32_t i = tid; i < total; i += THREADS) {
            const R rec = L.stage[i];
            const uint32_t bin = bin_of(rec.h);
            const uint32_t idx = i + L.delta[bin];       // 32-bit on purpose: delta is a wrapped difference
            if constexpr (PK) packed_store(out, seg0 + bin, cap2, idx, rec.h);
            else out[(seg0 + bin) * cap2 + idx] = rec;
        }
    } else {
#pragma unroll 2
        for (uint32_t i = tid; i < total; i += THREADS) {
            const R rec = L.stage[i];
            const uint32_t bin = bin_of(rec.h);
            const uint32_t idx = i + L.delta[bin];
            if (idx < cap2) {
                if constexpr (PK) packed_store(out, seg0 + bin, cap2, idx, rec.h);
                else out[(seg0 + bin) * cap2 + idx] = rec;
            }
            ovf_append(ovf, idx >= cap2, rec, n_overflow);
        }
    }
    n_overflow = (uint32_t)wave_sum(n_overflow);
    if (lane_id() == 0 && n_overflow) {
        atomicAdd(&ctr->n_overflow, (unsigned long long)n_overflow);
        atomicMax(&ctr->fail_mark, 0xFFFFFFFFULL - slab);     // the first slab that lost records (slab-wise accumulate: redone exactly)
    }
}

}  // namespace dk
