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
template <int THREADS, int PER_THREAD, int MIN_WAVES, class R, bool PK = false>
__global__ void __launch_bounds__(THREADS, MIN_WAVES)
repart_kernel(const R *__restrict__ in, const uint32_t *__restrict__ cnt1, uint32_t G, uint32_t capw,
              uint32_t tiles_per_piece, int b1, int b2, uint32_t cap2, R *__restrict__ out,
              uint32_t *__restrict__ cursor2, OvfList<R> ovf, Counters *ctr, int xcd_affine = 0, uint32_t bin_skew = 0,
              uint32_t bin0 = 0, uint32_t slab = 0)
{
    // bin0 / slab: slab-wise partition (bucketed_partition) -- this launch covers the level-1 bins from bin0 on, and `out`
    // holds the regions of these bins only (region (bin0 << b2) first); cursor2 is indexed by the region's global number
    constexpr int TILE = THREADS * PER_THREAD;
    constexpr int NB = THREADS >= MAX_BINS2 ? MAX_BINS2 : MAX_BINS;      // the bin scan is one thread per bin
    __shared__ SplitLds<THREADS, PER_THREAD, R, NB, false> L;
    const int tid = (int)threadIdx.x;
    // one-dimensional grid (the number of bins can exceed the 65535 of grid.y): bin-major, then piece, then tile
    const uint32_t per_bin = G * tiles_per_piece;
    uint32_t b = blockIdx.x / per_bin, bx = blockIdx.x % per_bin;
    if (xcd_affine) {
        // eight bins at a time, one per XCD (consecutive blocks are dealt round-robin over the XCDs): all tiles of a bin
        // then append to its 2^b2 frontiers through ONE L2, which assembles whole lines (speed only)
        const uint32_t slot = blockIdx.x >> 3;
        b = 8 * (slot / per_bin) + (blockIdx.x & 7);
        bx = slot % per_bin;
    }
    b += bin0;
    const uint32_t w = bx / tiles_per_piece, t0 = (bx % tiles_per_piece) * TILE;
    const uint64_t piece = (uint64_t)b * G + w;
    uint32_t n = cnt1[piece];
    if (n > capw) n = capw;
    if (t0 >= n) return;
    const int nbins = 1 << b2;
    const int shift = 64 - b1 - b2;
    auto bin_of = [=](uint64_t h) -> uint32_t { return (uint32_t)(h >> shift) & (uint32_t)(nbins - 1); };
    for (int i = tid; i < NB; i += THREADS) L.cnt[i] = 0;
    if (tid == 0) L.ovf_seen = 0;
    const R *src = in + piece * capw + (uint64_t)b * bin_skew;
    R hs[PER_THREAD];
#pragma unroll
    for (int j = 0; j < PER_THREAD; j++) {
        const uint32_t i = t0 + (uint32_t)j * THREADS + tid;
        hs[j] = src[i < n ? i : 0];
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
