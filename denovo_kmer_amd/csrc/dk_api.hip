// dk_api.hip -- C ABI of libdenovo_kmer.so (include/denovo_kmer.h): handle management, stream
// and workspace plumbing, kernel launches.  No CPU fallback exists: without a HIP device every
// entry point that needs one fails with DK_ERR_NO_DEVICE.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <new>

#include "dk_internal.h"
#include "dk_kernels_bucket.h"
#include "dk_comm.h"

static thread_local std::string g_create_err;

namespace dk {

dk_status fail(dk_engine *e, dk_status s, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (e) e->err = buf; else g_create_err = buf;
    return s;
}

// ---- device memory: reserved arenas first (dk_engine_reserve), then the caching pool (grow-only; blocks are reused
// across operations) ---------------
static uint64_t pool_in_use(const dk_engine *e)
{
    uint64_t n = 0;
    for (const auto &b : e->pool) if (b.in_use) n += b.bytes;
    for (const auto &a : e->arenas)
        for (const auto &g : a.segs) if (g.in_use) n += g.bytes;
    return n;
}

static void *arena_alloc(dk_engine *e, size_t bytes)
{
    // big blocks on 2-MiB boundaries (whole pages), small ones on 256 bytes
    const size_t align = bytes >= (8u << 20) ? (2u << 20) : 256;
    dk_arena *best_a = nullptr;
    size_t best_i = 0, best_bytes = ~(size_t)0, best_pad = 0;
    for (auto &a : e->arenas)
        for (size_t i = 0; i < a.segs.size(); i++) {
            const dk_arena_seg &g = a.segs[i];
            if (g.in_use) continue;
            const size_t start = ((uintptr_t)a.base + g.off + align - 1) / align * align - (uintptr_t)a.base;
            const size_t pad = start - g.off;
            if (g.bytes < pad + bytes || g.bytes >= best_bytes) continue;
            best_a = &a; best_i = i; best_bytes = g.bytes; best_pad = pad;
        }
    if (!best_a) return nullptr;
    std::vector<dk_arena_seg> &v = best_a->segs;
    if (best_pad) {                                          // the alignment gap stays a free segment of its own
        const dk_arena_seg g = v[best_i];
        v[best_i].bytes = best_pad;
        v.insert(v.begin() + best_i + 1, dk_arena_seg{g.off + best_pad, g.bytes - best_pad, false});
        best_i++;
    }
    if (v[best_i].bytes > bytes) {
        const dk_arena_seg g = v[best_i];
        v[best_i].bytes = bytes;
        v.insert(v.begin() + best_i + 1, dk_arena_seg{g.off + bytes, g.bytes - bytes, false});
    }
    v[best_i].in_use = true;
    return best_a->base + v[best_i].off;
}

static bool arena_free(dk_engine *e, void *p)
{
    for (auto &a : e->arenas) {
        if ((char *)p < a.base || (char *)p >= a.base + a.bytes) continue;
        std::vector<dk_arena_seg> &v = a.segs;
        const size_t off = (size_t)((char *)p - a.base);
        for (size_t i = 0; i < v.size(); i++) {
            if (v[i].off != off) continue;
            v[i].in_use = false;
            if (i + 1 < v.size() && !v[i + 1].in_use) { v[i].bytes += v[i + 1].bytes; v.erase(v.begin() + i + 1); }
            if (i > 0 && !v[i - 1].in_use) { v[i - 1].bytes += v[i].bytes; v.erase(v.begin() + i); }
            return true;
        }
        return true;                                         // inside the arena but not a segment start: ignore
    }
    return false;
}

dk_status pool_alloc(dk_engine *e, size_t bytes, void **out)
{
    if (bytes == 0) bytes = 256;
    bytes = (bytes + 255) & ~(size_t)255;
    if (void *p = arena_alloc(e, bytes)) {
        *out = p;
        e->pool_peak = std::max<uint64_t>(e->pool_peak, pool_in_use(e));
        return DK_OK;
    }
    int best = -1;
    for (size_t i = 0; i < e->pool.size(); i++) {
        dk_pool_block &b = e->pool[i];
        if (b.in_use || b.bytes < bytes || b.bytes > 2 * bytes + (1 << 20)) continue;
        if (best < 0 || b.bytes < e->pool[best].bytes) best = (int)i;
    }
    if (best >= 0) {
        e->pool[best].in_use = true;
        *out = e->pool[best].ptr;
        e->pool_peak = std::max<uint64_t>(e->pool_peak, pool_in_use(e));
        return DK_OK;
    }
    void *p = nullptr;
    hipError_t r = hipMalloc(&p, bytes);
    if (r != hipSuccess) {
        (void)hipGetLastError();     // the failure is handled here: do not leave it for the next hipGetLastError()
        // release every cached free block and retry once
        for (auto &b : e->pool)
            if (!b.in_use && b.ptr) { (void)hipFree(b.ptr); b.ptr = nullptr; b.bytes = 0; }
        e->pool.erase(std::remove_if(e->pool.begin(), e->pool.end(),
                                     [](const dk_pool_block &b) { return b.ptr == nullptr; }),
                      e->pool.end());
        r = hipMalloc(&p, bytes);
        if (r != hipSuccess) {
            (void)hipGetLastError();
            return fail(e, DK_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(r));
        }
    }
    e->pool.push_back({p, bytes, true});
    *out = p;
    e->pool_peak = std::max<uint64_t>(e->pool_peak, pool_in_use(e));
    return DK_OK;
}

void pool_free(dk_engine *e, void *p)
{
    if (!p) return;
    if (arena_free(e, p)) return;
    for (auto &b : e->pool)
        if (b.ptr == p) { b.in_use = false; return; }
}

void stage_begin(dk_engine *e)
{
    e->n_ev = 0;
    (void)hipEventRecord(e->ev[0], e->stream);
}

void stage_mark(dk_engine *e, const char *name)
{
    if (e->n_ev >= DK_MAX_MARKS) return;
    snprintf(e->ev_name[e->n_ev], sizeof e->ev_name[0], "%s", name);
    e->n_ev++;
    (void)hipEventRecord(e->ev[e->n_ev], e->stream);
}

dk_status stage_end(dk_engine *e)
{
    DK_HIP(e, hipStreamSynchronize(e->stream));
    dk_timings &t = e->timings;
    memset(&t, 0, sizeof t);
    // marks of the same name (the slabs of a slab-wise operation) are summed into one stage, in order of first appearance
    for (int i = 0; i < e->n_ev; i++) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e->ev[i], e->ev[i + 1]);
        uint32_t j = 0;
        while (j < t.n_stages && strncmp(t.stage_name[j], e->ev_name[i], sizeof t.stage_name[j]) != 0) j++;
        if (j == t.n_stages) {
            if (t.n_stages == DK_MAX_STAGES) continue;
            memcpy(t.stage_name[j], e->ev_name[i], sizeof t.stage_name[j]);
            t.n_stages++;
        }
        t.stage_ms[j] += ms;
    }
    if (e->n_ev) (void)hipEventElapsedTime(&t.total_ms, e->ev[0], e->ev[e->n_ev]);
#ifdef DK_STAMPS
    (void)hipMemcpy(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost);
    fprintf(stderr, "DK_STAMPS scan[count,scan,scatter,copy]=%llu %llu %llu %llu aux[4..7]=%llu %llu %llu %llu\n",
            e->h_ctr->dbg[0], e->h_ctr->dbg[1], e->h_ctr->dbg[2], e->h_ctr->dbg[3], e->h_ctr->dbg[4], e->h_ctr->dbg[5],
            e->h_ctr->dbg[6], e->h_ctr->dbg[7]);
#endif
    return DK_OK;
}

int grid_for(const dk_engine *e, uint64_t n_threads, int block)
{
    uint64_t blocks = (n_threads + block - 1) / block;
    const uint64_t cap = (uint64_t)e->n_cu * 8;     // 8 blocks of 256 per CU fill every wave slot
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static dk_status read_counters(dk_engine *e)
{
    DK_HIP(e, hipMemcpyAsync(e->h_ctr, e->d_ctr, sizeof(Counters), hipMemcpyDeviceToHost, e->stream));
    DK_HIP(e, hipStreamSynchronize(e->stream));
    return DK_OK;
}

static StreamView view_of(const dk_reads *r)
{
    StreamView s;
    s.bases = r->d_bases;
    s.mask = r->d_mask;
    s.n_bases = r->n_bases;
    s.n_bwords = (r->n_bases + 31) / 32;
    s.n_mwords = (r->n_bases + 63) / 64;
    return s;
}

// a batch that is still being uploaded (dk_reads_from_packed_async): the engine's stream waits for the copy stream
static dk_status reads_ready(dk_engine *e, const dk_reads *r)
{
    if (r->ready) DK_HIP(e, hipStreamWaitEvent(e->stream, r->ready, 0));
    return DK_OK;
}

// exact sets: number of 64-KiB segments = 2^T
static int exact_T_of(const dk_engine *e) { return (int)e->cfg.filter_log2_bits - 19; }

static FilterView fview_of(const dk_engine *e, const dk_set *s)
{
    FilterView f;
    f.words = s ? s->d_words : nullptr;
    f.log2_blocks = (int)e->cfg.filter_log2_bits - 9;
    f.n_hashes = (int)e->cfg.n_hashes;
    f.seed = e->cfg.seed;
    f.exact_T = s && s->exact ? exact_T_of(e) : 0;
    return f;
}

}  // namespace dk

using namespace dk;

static int ceil_log2(uint64_t v)
{
    int l = 0;
    while ((1ULL << l) < v) l++;
    return l;
}

template <bool WIDE>
static dk_status probe_direct(dk_engine *e, dk_set *s, const dk_reads *r, dk_result *res)
{
    const StreamView sv = view_of(r);
    const FilterView fv = fview_of(e, s);
    uint64_t cand_cap = std::min(r->n_bases, r->n_windows ? r->n_windows : r->n_bases);
    if (cand_cap == 0) cand_cap = 1;
    uint64_t *cand_lo = nullptr, *cand_hi = nullptr;
    uint32_t *slots = nullptr, *counts = nullptr;
    dk_status st = pool_alloc(e, cand_cap * 8, (void **)&cand_lo);
    if (st == DK_OK && WIDE) st = pool_alloc(e, cand_cap * 8, (void **)&cand_hi);
    auto cleanup = [&]() {
        pool_free(e, cand_lo);
        pool_free(e, cand_hi);
        pool_free(e, slots);
        pool_free(e, counts);
    };
    if (st != DK_OK) { cleanup(); return st; }

    probe_direct_kernel<WIDE><<<grid_for(e, r->n_bases, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
        sv, fv, (int)e->cfg.k, (int)e->cfg.canonical, e->d_ctr, cand_lo, cand_hi, cand_cap);
    hipError_t h = hipGetLastError();
    if (h != hipSuccess) { cleanup(); return fail(e, DK_ERR_HIP, "probe_direct launch failed: %s", hipGetErrorString(h)); }
    stage_mark(e, "probe_direct");
    st = read_counters(e);
    if (st != DK_OK) { cleanup(); return st; }
    const uint64_t n_cand = e->h_ctr->n_cand;
    if (n_cand > cand_cap) { cleanup(); return fail(e, DK_ERR_OVERFLOW, "candidate list overflow (%llu > %llu): n_windows under-stated?", (unsigned long long)n_cand, (unsigned long long)cand_cap); }
    if (n_cand >= 0xFFFFFFFFULL) { cleanup(); return fail(e, DK_ERR_OVERFLOW, "more than 2^32-1 absent k-mers in one batch; split the batch"); }
    if (n_cand == 0) { cleanup(); return DK_OK; }

    const int log2_cap = std::max(10, ceil_log2(2 * n_cand));
    const uint64_t cap = 1ULL << log2_cap;
    st = pool_alloc(e, cap * 4, (void **)&slots);
    if (st == DK_OK) st = pool_alloc(e, cap * 4, (void **)&counts);
    if (st == DK_OK) st = pool_alloc(e, n_cand * 8, (void **)&res->d_lo);
    if (st == DK_OK && WIDE) st = pool_alloc(e, n_cand * 8, (void **)&res->d_hi);
    if (st == DK_OK) st = pool_alloc(e, n_cand * 4, (void **)&res->d_cnt);
    if (st != DK_OK) { cleanup(); return st; }
    h = hipMemsetAsync(slots, 0xFF, cap * 4, e->stream);
    if (h == hipSuccess) h = hipMemsetAsync(counts, 0, cap * 4, e->stream);
    if (h == hipSuccess) {
        count_insert_kernel<WIDE><<<grid_for(e, n_cand, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
            cand_lo, cand_hi, n_cand, slots, counts, log2_cap, e->cfg.seed);
        h = hipGetLastError();
    }
    if (h == hipSuccess) {
        stage_mark(e, "count_insert");
        count_emit_kernel<WIDE><<<grid_for(e, cap, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
            cand_lo, cand_hi, slots, counts, cap, e->cfg.min_count, e->d_ctr, res->d_lo, res->d_hi, res->d_cnt);
        h = hipGetLastError();
    }
    if (h != hipSuccess) { cleanup(); return fail(e, DK_ERR_HIP, "count kernels failed: %s", hipGetErrorString(h)); }
    stage_mark(e, "count_emit");
    st = read_counters(e);
    cleanup();
    if (st != DK_OK) return st;
    res->n = e->h_ctr->n_emitted;
    res->n_regions = 1;
    res->region_cap = n_cand;
    res->region_n[0] = res->n;
    return DK_OK;
}

// Sum (k-mer, count) tables by k-mer.  The inputs are concatenated into dense candidate arrays; the index table is
// built and emitted one hash range at a time (at most 2^31 slots per pass), with 64-bit candidate indices once the
// inputs hold 2^32 - 1 entries or more, so the only bound is device memory (12 + 12 bytes per input entry plus
// the pass table).  An accumulator (dk_accum_*) is the better tool for a sample that arrives in many batches.
template <bool WIDE, class IdxT>
static dk_status merge_results(dk_engine *e, const dk_result *const *results, uint32_t n_results, uint32_t min_count,
                               dk_result *res, uint64_t total)
{
    uint64_t *lo = nullptr, *hi = nullptr;
    uint32_t *cnt = nullptr, *counts = nullptr;
    IdxT *slots = nullptr;
    auto cleanup = [&]() {
        pool_free(e, lo);
        pool_free(e, hi);
        pool_free(e, cnt);
        pool_free(e, slots);
        pool_free(e, counts);
    };
    int pbits = e->opt.merge_pass_bits;
    while (pbits < 16 && ((2 * total) >> pbits) > (1ULL << 31)) pbits++;
    dk_status st = pool_alloc(e, total * 8, (void **)&lo);
    if (st == DK_OK && WIDE) st = pool_alloc(e, total * 8, (void **)&hi);
    if (st == DK_OK) st = pool_alloc(e, total * 4, (void **)&cnt);
    if (st == DK_OK) st = pool_alloc(e, total * 8, (void **)&res->d_lo);
    if (st == DK_OK && WIDE) st = pool_alloc(e, total * 8, (void **)&res->d_hi);
    if (st == DK_OK) st = pool_alloc(e, total * 4, (void **)&res->d_cnt);
    if (st != DK_OK) { cleanup(); return st; }
    // concatenate the inputs (region by region) into dense candidate arrays
    hipError_t h = hipSuccess;
    uint64_t done = 0;
    for (uint32_t i = 0; i < n_results && h == hipSuccess; i++) {
        const dk_result *r = results[i];
        for (uint32_t g = 0; g < r->n_regions && h == hipSuccess; g++) {
            const uint64_t c = r->region_n[g], src = (uint64_t)g * r->region_cap;
            if (!c) continue;
            h = hipMemcpyAsync(lo + done, r->d_lo + src, c * 8, hipMemcpyDeviceToDevice, e->stream);
            if (h == hipSuccess && WIDE) h = hipMemcpyAsync(hi + done, r->d_hi + src, c * 8, hipMemcpyDeviceToDevice, e->stream);
            if (h == hipSuccess) h = hipMemcpyAsync(cnt + done, r->d_cnt + src, c * 4, hipMemcpyDeviceToDevice, e->stream);
            done += c;
        }
    }
    if (h != hipSuccess) { cleanup(); return fail(e, DK_ERR_HIP, "merge copies failed: %s", hipGetErrorString(h)); }
    // pass tables are sized for the mean share of a pass plus slack (the remixed hash ranges are uniform); should a pass
    // still not fit -- the kernel reports it instead of spinning -- the whole merge is redone with tables twice the size
    const uint64_t per_pass = (total >> pbits) + (pbits ? (total >> (pbits + 3)) + 65536 : 0);
    int log2_cap = std::max(e->opt.merge_undersize ? 4 : 10, ceil_log2(2 * per_pass) - e->opt.merge_undersize);   // (test hook: start too small)
    for (int attempt = 0;; attempt++) {
        const uint64_t cap = 1ULL << log2_cap;
        st = pool_alloc(e, cap * sizeof(IdxT), (void **)&slots);
        if (st == DK_OK) st = pool_alloc(e, cap * 4, (void **)&counts);
        if (st != DK_OK) { cleanup(); return st; }
        for (uint32_t q = 0; q < (1u << pbits) && h == hipSuccess; q++) {
            h = hipMemsetAsync(slots, 0xFF, cap * sizeof(IdxT), e->stream);
            if (h == hipSuccess) h = hipMemsetAsync(counts, 0, cap * 4, e->stream);
            if (h == hipSuccess) {
                merge_insert_kernel<WIDE, IdxT><<<grid_for(e, total, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                    lo, hi, cnt, total, slots, counts, log2_cap, e->cfg.seed, pbits, q, e->d_ctr);
                // entries are appended behind those of the earlier passes (Counters::n_emitted runs on)
                count_emit_kernel<WIDE, IdxT><<<grid_for(e, cap, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                    lo, hi, slots, counts, cap, min_count, e->d_ctr, res->d_lo, res->d_hi, res->d_cnt);
                h = hipGetLastError();
            }
        }
        if (h != hipSuccess) { cleanup(); return fail(e, DK_ERR_HIP, "merge kernels failed: %s", hipGetErrorString(h)); }
        stage_mark(e, attempt ? "merge_redo" : "merge");
        st = read_counters(e);
        if (st != DK_OK) { cleanup(); return st; }
        if (!e->h_ctr->n_overflow) break;
        if (attempt == 3) { cleanup(); return fail(e, DK_ERR_OVERFLOW, "merge: the pass tables overflowed four times over"); }
        pool_free(e, slots);
        pool_free(e, counts);
        slots = nullptr;
        counts = nullptr;
        log2_cap++;
        h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
        if (h != hipSuccess) { cleanup(); return fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h)); }
    }
    cleanup();
    res->n = e->h_ctr->n_emitted;
    res->n_regions = 1;
    res->region_cap = total;
    res->region_n[0] = res->n;
    return DK_OK;
}

template <bool WIDE>
static dk_status merge_results(dk_engine *e, const dk_result *const *results, uint32_t n_results, uint32_t min_count,
                               dk_result *res)
{
    uint64_t total = 0;
    for (uint32_t i = 0; i < n_results; i++) total += results[i]->n;
    if (total == 0) return DK_OK;
    if (total >= 0xFFFFFFFFULL || e->opt.merge_idx64) return merge_results<WIDE, uint64_t>(e, results, n_results, min_count, res, total);
    return merge_results<WIDE, uint32_t>(e, results, n_results, min_count, res, total);
}

// the exact redo path of dk_accum_add (and the small-batch path): the direct family lists the absent k-mers of the
// batch, which are hashed and appended to their units through global cursors
// h_from: only k-mers whose hash is >= h_from (what a slab-wise bucketed pass had not completed when its partition failed)
template <bool WIDE>
static dk_status accum_add_direct(dk_engine *e, dk_accum *a, const dk_reads *r, uint64_t h_from = 0)
{
    using R = typename RecOf<WIDE>::type;
    const StreamView sv = view_of(r);
    const FilterView fv = fview_of(e, a->s);
    uint64_t cand_cap = std::min(r->n_bases, r->n_windows ? r->n_windows : r->n_bases);
    if (cand_cap == 0) cand_cap = 1;
    uint64_t *cand_lo = nullptr, *cand_hi = nullptr;
    dk_status st = pool_alloc(e, cand_cap * 8, (void **)&cand_lo);
    if (st == DK_OK && WIDE) st = pool_alloc(e, cand_cap * 8, (void **)&cand_hi);
    hipError_t h = hipSuccess;
    if (st == DK_OK) {
        probe_direct_kernel<WIDE><<<grid_for(e, r->n_bases, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
            sv, fv, (int)e->cfg.k, (int)e->cfg.canonical, e->d_ctr, cand_lo, cand_hi, cand_cap);
        h = hipGetLastError();
        if (h == hipSuccess) {
            stage_mark(e, "probe_direct");
            st = read_counters(e);
        }
    }
    if (st == DK_OK && h == hipSuccess && e->h_ctr->n_cand > cand_cap)
        st = fail(e, DK_ERR_OVERFLOW, "candidate list overflow (%llu > %llu): n_windows under-stated?",
                  (unsigned long long)e->h_ctr->n_cand, (unsigned long long)cand_cap);
    if (st == DK_OK && h == hipSuccess && e->h_ctr->n_cand) {
        const uint64_t n = e->h_ctr->n_cand;
        if (a->packed && !WIDE)
            acc_append_kmers_kernel<WIDE, !WIDE><<<grid_for(e, n, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                cand_lo, cand_hi, n, e->cfg.seed, a->wbits, a->widx, accum_unit_bits(a), accum_unit_base(a), accum_out<R>(e, a), e->d_ctr, h_from);
        else
            acc_append_kmers_kernel<WIDE, false><<<grid_for(e, n, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                cand_lo, cand_hi, n, e->cfg.seed, a->wbits, a->widx, accum_unit_bits(a), accum_unit_base(a), accum_out<R>(e, a), e->d_ctr, h_from);
        h = hipGetLastError();
        if (h == hipSuccess) stage_mark(e, "acc_append");
    }
    pool_free(e, cand_lo);
    pool_free(e, cand_hi);
    if (st == DK_OK && h != hipSuccess) st = fail(e, DK_ERR_HIP, "direct accumulate failed: %s", hipGetErrorString(h));
    if (st == DK_OK) {
        e->h_ctr->n_absent = 0;                                 // the direct kernel tallied every absent window, the append only those inside the window
        h = hipMemsetAsync(&e->d_ctr->n_absent, 0, 8, e->stream);
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "counter reset failed: %s", hipGetErrorString(h));
    }
    if (st == DK_OK) st = sync_counters(e, "direct accumulate");
    return st;
}

// Count an accumulator's units.  Single GPU: its own store.  Multi-GPU (pieces != nullptr): the units
// [first_unit, first_unit + n_units) of the window from n_pieces slices in piece-major order -- what the ranks sent
// this rank for its share of the hash space -- plus `extra`, the gathered overflow lists (records of other ranks'
// shares are skipped).
template <bool WIDE>
static dk_status accum_finish_t(dk_engine *e, dk_accum *a, uint32_t min_count, dk_result *res, const void *pieces,
                                const uint32_t *piece_fills, uint32_t n_pieces, uint64_t first_unit, uint64_t n_units,
                                const void *extra_in, uint64_t n_extra_in, uint64_t *n_records_out)
{
    using R = typename RecOf<WIDE>::type;
    const int Tu = accum_unit_bits(a);
    const uint64_t unit_base = accum_unit_base(a) + first_unit;
    PieceList<R> list{(const R *)a->store, a->fill, 1, a->unit_cap, nullptr, nullptr};
    const R *ovf_recs = (const R *)a->ovf;
    unsigned long long n_aovf = 0;
    uint64_t n_records = a->n_absent;
    if (pieces) {
        list = PieceList<R>{(const R *)pieces, piece_fills, n_pieces, a->unit_cap, nullptr, nullptr};
        list.n_segs = n_units;
        ovf_recs = (const R *)extra_in;
        n_aovf = n_extra_in;
        // the records these pieces hold (sizes the table)
        DK_HIP(e, hipMemsetAsync(&e->d_ctr->dbg[0], 0, 8, e->stream));
        fill_sum_kernel<<<grid_for(e, n_pieces * n_units, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
            piece_fills, n_pieces * n_units, a->unit_cap, &e->d_ctr->dbg[0]);
        DK_HIP(e, hipGetLastError());
        unsigned long long sum = 0;
        DK_HIP(e, hipMemcpyAsync(&sum, &e->d_ctr->dbg[0], 8, hipMemcpyDeviceToHost, e->stream));
        DK_HIP(e, hipStreamSynchronize(e->stream));
        n_records = sum + n_aovf;
    } else {
        // occurrences that found their unit full: sorted by unit (CSR) and counted with it
        DK_HIP(e, hipMemcpyAsync(&n_aovf, a->d_novf, 8, hipMemcpyDeviceToHost, e->stream));
        DK_HIP(e, hipStreamSynchronize(e->stream));
        if (n_aovf > a->ovf_cap) n_aovf = a->ovf_cap;
    }
    if (n_records_out) *n_records_out = n_records;
    R *extra = nullptr;
    uint32_t *idx = nullptr;
    dk_status st = DK_OK;
    if (n_aovf) {
        st = pool_alloc(e, n_aovf * sizeof(R), (void **)&extra);
        if (st == DK_OK) st = pool_alloc(e, (3 * n_units + 1) * 4, (void **)&idx);
        hipError_t h = hipSuccess;
        if (st == DK_OK) {
            uint32_t *hist = idx, *off = hist + n_units, *fill = off + n_units + 1;
            h = hipMemsetAsync(idx, 0, (3 * n_units + 1) * 4, e->stream);
            if (h == hipSuccess) {
                unit_hist_kernel<R><<<grid_for(e, n_aovf, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(ovf_recs, n_aovf, Tu, unit_base, hist, n_units);
                ovf_scan_kernel<<<1, 1024, 0, e->stream>>>(hist, off, (uint32_t)n_units);
                ovf_scatter_kernel<R><<<grid_for(e, n_aovf, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
                    ovf_recs, n_aovf, Tu, unit_base, off, fill, extra, n_units);
                h = hipGetLastError();
            }
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "accumulator overflow sort failed: %s", hipGetErrorString(h));
            else stage_mark(e, "acc_ovf_sort");
            list.extra = extra;
            list.extra_off = off;
        }
    }
    if (st == DK_OK)
        st = bucketed_count_stage<WIDE>(e, list, n_units, Tu, unit_base, n_records, n_aovf, min_count, res,
                                        (uint64_t)(pieces ? n_pieces : 1u) * n_units * a->unit_cap, a->packed);
    pool_free(e, extra);
    pool_free(e, idx);
    return st;
}

#define CHECK_ARG(e, cond)                                                              \
    do {                                                                                \
        if (!(cond)) return dk::fail((e), DK_ERR_INVALID_ARG, "invalid argument: %s", #cond); \
    } while (0)

extern "C" {

int32_t dk_abi_version(void) { return DK_ABI_VERSION; }

const char *dk_status_string(dk_status s)
{
    switch (s) {
    case DK_OK: return "ok";
    case DK_ERR_INVALID_ARG: return "invalid argument";
    case DK_ERR_NO_DEVICE: return "no HIP device";
    case DK_ERR_HIP: return "HIP runtime error";
    case DK_ERR_OOM: return "out of device memory";
    case DK_ERR_UNSUPPORTED: return "unsupported configuration";
    case DK_ERR_OVERFLOW: return "capacity overflow";
    case DK_ERR_SET_FULL: return "exact set full";
    default: return "unknown status";
    }
}

// ---- engine ------------------------------------------------------------------------------------
dk_status dk_engine_create(const dk_config *cfg, dk_engine **out)
{
    if (!cfg || !out) return fail(nullptr, DK_ERR_INVALID_ARG, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->struct_size != sizeof(dk_config))
        return fail(nullptr, DK_ERR_INVALID_ARG, "dk_config.struct_size %llu != %zu",
                    (unsigned long long)cfg->struct_size, sizeof(dk_config));
    if (cfg->k < 1 || cfg->k > 64) return fail(nullptr, DK_ERR_INVALID_ARG, "k=%u outside 1..64", cfg->k);
    if (cfg->filter_log2_bits < 20 || cfg->filter_log2_bits > 40)
        return fail(nullptr, DK_ERR_INVALID_ARG, "filter_log2_bits=%u outside 20..40", cfg->filter_log2_bits);
    if (cfg->n_hashes < 1 || cfg->n_hashes > 16)
        return fail(nullptr, DK_ERR_INVALID_ARG, "n_hashes=%u outside 1..16", cfg->n_hashes);
    if (cfg->min_count < 1) return fail(nullptr, DK_ERR_INVALID_ARG, "min_count must be >= 1");
    if (cfg->mode > DK_MODE_BUCKETED) return fail(nullptr, DK_ERR_INVALID_ARG, "unknown mode %u", cfg->mode);
    if (cfg->set_kind > DK_SET_EXACT) return fail(nullptr, DK_ERR_INVALID_ARG, "unknown set_kind %u", cfg->set_kind);

    int n_dev = 0;
    hipError_t r = hipGetDeviceCount(&n_dev);
    if (r != hipSuccess || n_dev == 0)
        return fail(nullptr, DK_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU path",
                    r == hipSuccess ? "device count 0" : hipGetErrorString(r));
    if (cfg->device_id < 0 || cfg->device_id >= n_dev)
        return fail(nullptr, DK_ERR_INVALID_ARG, "device_id %d outside 0..%d", cfg->device_id, n_dev - 1);

    dk_engine *e = new (std::nothrow) dk_engine();
    if (!e) return fail(nullptr, DK_ERR_OOM, "host allocation failed");
    e->cfg = *cfg;
    e->device = cfg->device_id;
    e->own_stream = cfg->stream == nullptr;
    e->d_ctr = nullptr;
    e->h_ctr = nullptr;
    e->n_ev = 0;
    memset(&e->timings, 0, sizeof e->timings);
    for (auto &ev : e->ev) ev = nullptr;

    auto bail = [&](const char *what, hipError_t err) {
        dk_status s = fail(nullptr, DK_ERR_HIP, "%s failed: %s", what, hipGetErrorString(err));
        dk_engine_destroy(e);
        return s;
    };
    if ((r = hipSetDevice(e->device)) != hipSuccess) return bail("hipSetDevice", r);
    hipDeviceProp_t prop;
    if ((r = hipGetDeviceProperties(&prop, e->device)) != hipSuccess) return bail("hipGetDeviceProperties", r);
    e->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (e->own_stream) {
        if ((r = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess)
            return bail("hipStreamCreate", r);
    } else {
        e->stream = (hipStream_t)cfg->stream;
    }
    if ((r = hipMalloc((void **)&e->d_ctr, sizeof(Counters))) != hipSuccess) return bail("hipMalloc", r);
    if ((r = hipHostMalloc((void **)&e->h_ctr, sizeof(Counters), hipHostMallocDefault)) != hipSuccess)
        return bail("hipHostMalloc", r);
    for (auto &ev : e->ev)
        if ((r = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", r);
    *out = e;
    return DK_OK;
}

void dk_engine_destroy(dk_engine *e)
{
    if (!e) return;
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    (void)dk_comm_finalize(e);
    for (auto &b : e->pool)
        if (b.ptr) (void)hipFree(b.ptr);
    for (auto &a : e->arenas)
        if (a.base) (void)hipFree(a.base);
    if (e->d_ctr) (void)hipFree(e->d_ctr);
    if (e->h_ctr) (void)hipHostFree(e->h_ctr);
    for (auto &ev : e->ev)
        if (ev) (void)hipEventDestroy(ev);
    if (e->copy_stream) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamDestroy(e->copy_stream); }
    if (e->copy_ev) (void)hipEventDestroy(e->copy_ev);
    if (e->own_stream && e->stream) (void)hipStreamDestroy(e->stream);
    delete e;
}

const char *dk_last_error(const dk_engine *e) { return e ? e->err.c_str() : g_create_err.c_str(); }

dk_status dk_engine_synchronize(dk_engine *e)
{
    if (!e) return DK_ERR_INVALID_ARG;
    DK_HIP(e, hipSetDevice(e->device));
    DK_HIP(e, hipStreamSynchronize(e->stream));
    return DK_OK;
}

dk_status dk_engine_timings(const dk_engine *e, dk_timings *out)
{
    if (!e || !out) return DK_ERR_INVALID_ARG;
    *out = e->timings;
    return DK_OK;
}

dk_status dk_engine_config(const dk_engine *e, dk_config *out)
{
    if (!e || !out) return DK_ERR_INVALID_ARG;
    *out = e->cfg;
    return DK_OK;
}

dk_status dk_engine_trim(dk_engine *e, uint64_t *bytes_freed)
{
    if (!e) return DK_ERR_INVALID_ARG;
    DK_HIP(e, hipSetDevice(e->device));
    DK_HIP(e, hipStreamSynchronize(e->stream));
    uint64_t freed = 0;
    for (auto &b : e->pool)
        if (!b.in_use && b.ptr) {
            (void)hipFree(b.ptr);
            freed += b.bytes;
            b.ptr = nullptr;
            b.bytes = 0;
        }
    e->pool.erase(std::remove_if(e->pool.begin(), e->pool.end(), [](const dk_pool_block &b) { return b.ptr == nullptr; }),
                  e->pool.end());
    if (bytes_freed) *bytes_freed = freed;
    return DK_OK;
}

dk_status dk_engine_reserve(dk_engine *e, uint64_t bytes)
{
    if (!e) return DK_ERR_INVALID_ARG;
    DK_HIP(e, hipSetDevice(e->device));
    if (bytes == 0) {
        // hand back every arena nothing lives in
        DK_HIP(e, hipStreamSynchronize(e->stream));
        for (auto &a : e->arenas)
            if (a.segs.size() == 1 && !a.segs[0].in_use) { (void)hipFree(a.base); a.base = nullptr; }
        e->arenas.erase(std::remove_if(e->arenas.begin(), e->arenas.end(), [](const dk_arena &a) { return a.base == nullptr; }),
                        e->arenas.end());
        return DK_OK;
    }
    bytes = (bytes + (2u << 20) - 1) & ~(uint64_t)((2u << 20) - 1);
    void *p = nullptr;
    const hipError_t r = hipMalloc(&p, bytes);
    if (r != hipSuccess) {
        (void)hipGetLastError();
        return fail(e, DK_ERR_OOM, "hipMalloc(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(r));
    }
    dk_arena a;
    a.base = (char *)p;
    a.bytes = bytes;
    a.segs.push_back(dk_arena_seg{0, (size_t)bytes, false});
    e->arenas.push_back(a);
    return DK_OK;
}

dk_status dk_engine_get_info(const dk_engine *e, const char *name, int64_t *value)
{
    if (!e || !name || !value) return DK_ERR_INVALID_ARG;
    const dk_plan_info &p = e->plan;
    uint64_t cached = 0, reserved = 0;
    for (const auto &b : e->pool) cached += b.bytes;
    for (const auto &a : e->arenas) reserved += a.bytes;
    const struct { const char *name; int64_t v; } info[] = {
        {"plan_levels", p.levels}, {"plan_b1", p.b1}, {"plan_b2", p.b2}, {"plan_b3", p.b3}, {"plan_sbits", p.sbits},
        {"plan_slabs", p.slabs}, {"plan_scan_variant", p.scan_variant}, {"plan_segment_bits", p.T},
        {"pool_bytes_in_use", (int64_t)pool_in_use(e)}, {"pool_bytes_cached", (int64_t)cached},
        {"pool_bytes_reserved", (int64_t)reserved}, {"pool_bytes_peak", (int64_t)e->pool_peak},
    };
    for (const auto &i : info)
        if (strcmp(name, i.name) == 0) { *value = i.v; return DK_OK; }
#ifdef DK_DEBUG_INFO
    if (strncmp(name, "dbg", 3) == 0 && name[3] >= '0' && name[3] <= '7') {       // diagnostic builds: Counters::dbg[n] as the device holds it
        Counters c;
        (void)hipMemcpy(&c, e->d_ctr, sizeof c, hipMemcpyDeviceToHost);
        *value = (int64_t)c.dbg[name[3] - '0'];
        return DK_OK;
    }
#endif
    return DK_ERR_INVALID_ARG;
}

dk_status dk_engine_set_option(dk_engine *e, const char *name, int64_t value)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, name != nullptr);
    struct Opt { const char *name; int dk_options::*field; int64_t lo, hi; };
    static const Opt opts[] = {
        {"multiplicity_hint", &dk_options::multiplicity_hint, 0, 1 << 20},
        {"scan_variant", &dk_options::scan_variant, 0, 6},
        {"repart_variant", &dk_options::repart_variant, 0, 1},
        {"force_l3", &dk_options::force_l3, 0, 1},
        {"b1_up", &dk_options::b1_up, -4, 4},
        {"count_seg", &dk_options::count_seg, 0, 1 << 30},
        {"cnt_mid", &dk_options::cnt_mid, 0, 1 << 30},
        {"cnt_big", &dk_options::cnt_big, 0, 1 << 30},
        {"cnt_split_to", &dk_options::cnt_split_to, 0, 1 << 30},
        {"sub_split", &dk_options::sub_split, 0, 9},
        {"repart_plain", &dk_options::repart_plain, 0, 1},
        {"repart_pieces", &dk_options::repart_pieces, 0, 2},
        {"repart_bits", &dk_options::repart_bits, 0, 10},
        {"scan_bits", &dk_options::scan_bits, 0, 10},
        {"slabs", &dk_options::slabs, 0, 1024},
        {"slab_mb", &dk_options::slab_mb, 0, 1 << 20},
        {"ovf_cap", &dk_options::ovf_cap, 0, 1 << 30},
        {"accum_plain", &dk_options::accum_plain, 0, 1},
        {"l2_packed", &dk_options::l2_packed, 0, 1},
        {"accum_min_u", &dk_options::accum_min_u, 0, 10},
        {"mode", &dk_options::mode, 0, 2},
        {"kmers_plain", &dk_options::kmers_plain, 0, 1},
        {"scan_positions", &dk_options::scan_positions, 0, 1},
        {"merge_undersize", &dk_options::merge_undersize, 0, 10},
        {"comm_staging_kb", &dk_options::comm_staging_kb, 0, 1 << 30},
        {"l1_layout", &dk_options::l1_layout, 0, 1},
        {"l1_skew", &dk_options::l1_skew, 0, 1 << 24},
        {"merge_pass_bits", &dk_options::merge_pass_bits, 0, 8},
        {"sink_plain", &dk_options::sink_plain, 0, 1},
        {"accum_unit_cap", &dk_options::accum_unit_cap, 0, 1 << 20},
        {"merge_idx64", &dk_options::merge_idx64, 0, 1},
    };
    for (const Opt &o : opts) {
        if (strcmp(name, o.name) != 0) continue;
        if (value < o.lo || value > o.hi)
            return fail(e, DK_ERR_INVALID_ARG, "option %s: %lld outside %lld..%lld", name, (long long)value, (long long)o.lo, (long long)o.hi);
        if (!strcmp(name, "count_seg") && value != 0 && value < 64)
            return fail(e, DK_ERR_INVALID_ARG, "option count_seg: %lld below 64", (long long)value);
        e->opt.*(o.field) = (int)value;
        return DK_OK;
    }
    return fail(e, DK_ERR_INVALID_ARG, "unknown option \"%s\"", name);
}

// ---- read batches -----------------------------------------------------------------------------
static dk_status reads_alloc(dk_engine *e, uint64_t n_bases, uint64_t n_reads, uint64_t n_windows, dk_reads **out)
{
    dk_reads *r = new (std::nothrow) dk_reads();
    if (!r) return fail(e, DK_ERR_OOM, "host allocation failed");
    r->e = e;
    r->n_bases = n_bases;
    r->n_reads = n_reads;
    r->n_windows = n_windows;
    r->owns = true;
    r->d_bases = r->d_mask = nullptr;
    const size_t bw = (n_bases + 31) / 32, mw = (n_bases + 63) / 64;
    dk_status s = pool_alloc(e, (bw + 2) * 8, (void **)&r->d_bases);
    if (s == DK_OK) s = pool_alloc(e, (mw + 2) * 8, (void **)&r->d_mask);
    if (s == DK_OK) s = pool_alloc(e, 256, (void **)&r->d_uniform);
    if (s != DK_OK) { dk_reads_destroy(r); return s; }
    *out = r;
    return DK_OK;
}

// The batch may consist of reads of one length (n_bases = n_reads * (L + 1)): have the device check it on `stream`, behind the
// copies that bring the mask.  known = true: the caller built the stream itself (synthetic reads, uniform ASCII reads).
static hipError_t reads_mark_uniform(dk_reads *r, hipStream_t stream, bool known)
{
    r->stride = 0;
    if (!r->d_uniform || r->n_reads == 0 || r->n_bases % r->n_reads != 0) return hipSuccess;
    const uint64_t stride = r->n_bases / r->n_reads;
    if (stride < 2 || stride > 0xFFFFFFFFULL || r->n_reads > 0x7FFFFFFFULL) return hipSuccess;
    hipError_t h = hipMemsetD32Async((hipDeviceptr_t)r->d_uniform, 1, 1, stream);
    if (h == hipSuccess && !known) {
        const uint64_t grid = (r->n_reads + DIRECT_BLOCK - 1) / DIRECT_BLOCK;
        verify_stride_kernel<<<(unsigned)grid, DIRECT_BLOCK, 0, stream>>>(r->d_mask, r->n_reads, (uint32_t)stride, r->d_uniform);
        h = hipGetLastError();
    }
    if (h == hipSuccess) r->stride = (uint32_t)stride;
    return h;
}

static uint64_t windows_of(const uint64_t *offsets, uint64_t n_reads, uint32_t k)
{
    uint64_t w = 0;
    for (uint64_t i = 0; i < n_reads; i++) {
        const uint64_t l = offsets[i + 1] - offsets[i];
        if (l >= k) w += l - k + 1;
    }
    return w;
}

dk_status dk_reads_from_ascii(dk_engine *e, const uint8_t *seq, const uint64_t *offsets,
                              uint64_t n_reads, dk_reads **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr);
    CHECK_ARG(e, offsets != nullptr || n_reads == 0);
    *out = nullptr;
    DK_HIP(e, hipSetDevice(e->device));
    CHECK_ARG(e, n_reads == 0 || offsets[0] == 0);
    for (uint64_t i = 0; i < n_reads; i++) CHECK_ARG(e, offsets[i + 1] >= offsets[i]);
    const uint64_t n_seq = n_reads ? offsets[n_reads] : 0;
    CHECK_ARG(e, seq != nullptr || n_seq == 0);
    const uint64_t n_bases = n_seq + n_reads;
    dk_reads *r = nullptr;
    DK_TRY(reads_alloc(e, n_bases, n_reads, windows_of(offsets, n_reads, e->cfg.k), &r));
    if (n_bases) {
        uint8_t *d_seq = nullptr;
        uint64_t *d_off = nullptr;
        dk_status s = pool_alloc(e, n_seq + 16, (void **)&d_seq);
        if (s == DK_OK) s = pool_alloc(e, (n_reads + 1) * 8, (void **)&d_off);
        if (s != DK_OK) { pool_free(e, d_seq); dk_reads_destroy(r); return s; }
        hipError_t h = hipSuccess;
        if (n_seq) h = hipMemcpyAsync(d_seq, seq, n_seq, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess) h = hipMemcpyAsync(d_off, offsets, (n_reads + 1) * 8, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess) {
            const uint64_t chunks = (n_bases + 63) / 64;
            const int grid = (int)((chunks + DIRECT_BLOCK - 1) / DIRECT_BLOCK);
            pack_ascii_kernel<<<grid, DIRECT_BLOCK, 0, e->stream>>>(d_seq, d_off, n_reads, n_bases, r->d_bases, r->d_mask);
            h = hipGetLastError();
        }
        if (h == hipSuccess) {
            bool uniform = true;                               // reads of one length: the scan deals windows, not positions, to its threads
            for (uint64_t i = 1; i < n_reads && uniform; i++) uniform = offsets[i + 1] - offsets[i] == offsets[1] - offsets[0];
            if (uniform) h = reads_mark_uniform(r, e->stream, true);
        }
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);   // seq/offsets are borrowed only for the call
        pool_free(e, d_seq);
        pool_free(e, d_off);
        if (h != hipSuccess) {
            dk_reads_destroy(r);
            return fail(e, DK_ERR_HIP, "packing reads failed: %s", hipGetErrorString(h));
        }
    }
    *out = r;
    return DK_OK;
}

dk_status dk_reads_from_packed(dk_engine *e, const uint64_t *bases, const uint64_t *mask,
                               uint64_t n_bases, uint64_t n_reads, uint64_t n_windows, dk_reads **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr);
    CHECK_ARG(e, (bases != nullptr && mask != nullptr) || n_bases == 0);
    *out = nullptr;
    // the stream must end with a separator (the kernels judge a window that would run past the end by its flags)
    if (n_bases && !((mask[(n_bases - 1) >> 6] >> (63 - ((n_bases - 1) & 63))) & 1ULL))
        return fail(e, DK_ERR_INVALID_ARG, "packed stream does not end with a flagged separator position");
    DK_HIP(e, hipSetDevice(e->device));
    dk_reads *r = nullptr;
    DK_TRY(reads_alloc(e, n_bases, n_reads, n_windows, &r));
    if (n_bases) {
        hipError_t h = hipMemcpyAsync(r->d_bases, bases, (n_bases + 31) / 32 * 8, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess) h = hipMemcpyAsync(r->d_mask, mask, (n_bases + 63) / 64 * 8, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess) h = reads_mark_uniform(r, e->stream, false);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) {
            dk_reads_destroy(r);
            return fail(e, DK_ERR_HIP, "uploading packed reads failed: %s", hipGetErrorString(h));
        }
    }
    *out = r;
    return DK_OK;
}

// Overlapped ingest: the copies run on the engine's copy stream and the call returns at once; every operation that
// consumes the batch makes the engine's stream wait for them, so batch i + 1 travels over PCIe while batch i is being
// probed.  The host buffers must stay valid and unchanged until dk_reads_wait returns; they should be pinned
// (dk_host_alloc), or the runtime stages the copy and the call blocks.
dk_status dk_reads_from_packed_async(dk_engine *e, const uint64_t *bases, const uint64_t *mask,
                                     uint64_t n_bases, uint64_t n_reads, uint64_t n_windows, dk_reads **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr);
    CHECK_ARG(e, (bases != nullptr && mask != nullptr) || n_bases == 0);
    *out = nullptr;
    if (n_bases && !((mask[(n_bases - 1) >> 6] >> (63 - ((n_bases - 1) & 63))) & 1ULL))
        return fail(e, DK_ERR_INVALID_ARG, "packed stream does not end with a flagged separator position");
    DK_HIP(e, hipSetDevice(e->device));
    if (!e->copy_stream) DK_HIP(e, hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    if (!e->copy_ev) DK_HIP(e, hipEventCreateWithFlags(&e->copy_ev, hipEventDisableTiming));
    dk_reads *r = nullptr;
    DK_TRY(reads_alloc(e, n_bases, n_reads, n_windows, &r));
    hipError_t h = hipEventCreateWithFlags(&r->ready, hipEventDisableTiming);
    if (h == hipSuccess && n_bases) {
        // the block may have served an operation that is still running on the engine's stream: copy after it
        h = hipEventRecord(e->copy_ev, e->stream);
        if (h == hipSuccess) h = hipStreamWaitEvent(e->copy_stream, e->copy_ev, 0);
        if (h == hipSuccess) h = hipMemcpyAsync(r->d_bases, bases, (n_bases + 31) / 32 * 8, hipMemcpyHostToDevice, e->copy_stream);
        if (h == hipSuccess) h = hipMemcpyAsync(r->d_mask, mask, (n_bases + 63) / 64 * 8, hipMemcpyHostToDevice, e->copy_stream);
        if (h == hipSuccess) h = reads_mark_uniform(r, e->copy_stream, false);
    }
    if (h == hipSuccess) h = hipEventRecord(r->ready, e->copy_stream);
    if (h != hipSuccess) {
        dk_reads_destroy(r);
        return fail(e, DK_ERR_HIP, "uploading packed reads failed: %s", hipGetErrorString(h));
    }
    *out = r;
    return DK_OK;
}

dk_status dk_reads_wait(dk_reads *r)
{
    if (!r) return DK_ERR_INVALID_ARG;
    if (r->ready) DK_HIP(r->e, hipEventSynchronize(r->ready));
    return DK_OK;
}

dk_status dk_host_alloc(uint64_t bytes, void **out)
{
    if (!out) return DK_ERR_INVALID_ARG;
    *out = nullptr;
    const hipError_t h = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (h != hipSuccess) { (void)hipGetLastError(); return fail(nullptr, DK_ERR_OOM, "hipHostMalloc(%llu bytes) failed: %s", (unsigned long long)bytes, hipGetErrorString(h)); }
    return DK_OK;
}

void dk_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

dk_status dk_reads_attach_device(dk_engine *e, const void *d_bases, const void *d_mask,
                                 uint64_t n_bases, uint64_t n_reads, uint64_t n_windows, dk_reads **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr);
    CHECK_ARG(e, (d_bases != nullptr && d_mask != nullptr) || n_bases == 0);
    if (n_bases) {
        // same requirement as dk_reads_from_packed: one word of the caller's mask is read back to check it
        uint64_t last = 0;
        DK_HIP(e, hipSetDevice(e->device));
        DK_HIP(e, hipMemcpy(&last, (const uint64_t *)d_mask + ((n_bases - 1) >> 6), 8, hipMemcpyDeviceToHost));
        if (!((last >> (63 - ((n_bases - 1) & 63))) & 1ULL))
            return fail(e, DK_ERR_INVALID_ARG, "packed stream does not end with a flagged separator position");
    }
    dk_reads *r = new (std::nothrow) dk_reads();
    if (!r) return fail(e, DK_ERR_OOM, "host allocation failed");
    r->e = e;
    r->d_bases = (uint64_t *)d_bases;
    r->d_mask = (uint64_t *)d_mask;
    r->n_bases = n_bases;
    r->n_reads = n_reads;
    r->n_windows = n_windows;
    r->owns = false;
    if (n_bases && pool_alloc(e, 256, (void **)&r->d_uniform) == DK_OK) {
        const hipError_t h = reads_mark_uniform(r, e->stream, false);
        if (h != hipSuccess) { dk_reads_destroy(r); return fail(e, DK_ERR_HIP, "checking the read lengths failed: %s", hipGetErrorString(h)); }
    }
    *out = r;
    return DK_OK;
}

dk_status dk_reads_synth(dk_engine *e, const dk_synth_config *cfg, int32_t sample,
                         uint64_t first_read, uint64_t n_reads, dk_reads **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, cfg != nullptr && out != nullptr);
    CHECK_ARG(e, cfg->struct_size == sizeof(dk_synth_config));
    CHECK_ARG(e, sample >= 0 && sample <= 2);
    CHECK_ARG(e, cfg->read_len >= 1 && cfg->genome_len >= cfg->read_len);
    CHECK_ARG(e, cfg->xover_log2 < 64);
    *out = nullptr;
    DK_HIP(e, hipSetDevice(e->device));
    const uint64_t L = cfg->read_len;
    const uint64_t n_bases = n_reads * (L + 1);
    const uint64_t n_windows = L >= e->cfg.k ? n_reads * (L - e->cfg.k + 1) : 0;
    dk_reads *r = nullptr;
    DK_TRY(reads_alloc(e, n_bases, n_reads, n_windows, &r));
    if (n_bases) {
        SynthParams p;
        p.seed = cfg->seed;
        p.genome_len = cfg->genome_len;
        p.span = cfg->genome_len - L + 1;
        p.read_len = cfg->read_len;
        p.xover_log2 = cfg->xover_log2;
        p.snv_thr = cfg->snv_thr;
        p.denovo_thr = cfg->denovo_thr;
        p.err_thr = cfg->err_thr;
        p.n_thr = cfg->n_thr;
        p.sample = sample;
        p.first_read = first_read;
        p.n_reads = n_reads;
        const uint64_t chunks = (n_bases + 63) / 64;
        const uint64_t grid = (chunks + DIRECT_BLOCK - 1) / DIRECT_BLOCK;
        if (grid > 0x7FFFFFFFULL) { dk_reads_destroy(r); return fail(e, DK_ERR_INVALID_ARG, "synthetic batch too large"); }
        synth_kernel<<<(int)grid, DIRECT_BLOCK, 0, e->stream>>>(p, n_bases, r->d_bases, r->d_mask);
        hipError_t h = hipGetLastError();
        if (h == hipSuccess) h = reads_mark_uniform(r, e->stream, true);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) {
            dk_reads_destroy(r);
            return fail(e, DK_ERR_HIP, "synthetic read generation failed: %s", hipGetErrorString(h));
        }
    }
    *out = r;
    return DK_OK;
}

dk_status dk_reads_stats(const dk_reads *r, dk_stats *out)
{
    if (!r || !out) return DK_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->n_reads = r->n_reads;
    out->n_bases = r->n_bases;
    out->n_windows = r->n_windows;
    return DK_OK;
}

dk_status dk_reads_download(const dk_reads *r, uint64_t *bases, uint64_t *mask)
{
    if (!r) return DK_ERR_INVALID_ARG;
    dk_engine *e = r->e;
    CHECK_ARG(e, (bases && mask) || r->n_bases == 0);
    DK_HIP(e, hipSetDevice(e->device));
    DK_TRY(reads_ready(e, r));
    if (r->n_bases) {
        DK_HIP(e, hipMemcpyAsync(bases, r->d_bases, (r->n_bases + 31) / 32 * 8, hipMemcpyDeviceToHost, e->stream));
        DK_HIP(e, hipMemcpyAsync(mask, r->d_mask, (r->n_bases + 63) / 64 * 8, hipMemcpyDeviceToHost, e->stream));
        DK_HIP(e, hipStreamSynchronize(e->stream));
    }
    return DK_OK;
}

static bool is_device_ptr(const void *p)
{
    hipPointerAttribute_t attr;
    const hipError_t h = hipPointerGetAttributes(&attr, p);
    if (h != hipSuccess) (void)hipGetLastError();          // plain host memory is reported as an error: not one
    return h == hipSuccess && attr.type == hipMemoryTypeDevice;
}

// device pointer as is; host pointer -> pool buffer that is copied back by finish()
struct OutBuf {
    uint64_t *user = nullptr, *dev = nullptr;
    bool staged = false;
};

static dk_status out_prepare(dk_engine *e, uint64_t *user, size_t bytes, OutBuf *b)
{
    b->user = user;
    if (!user || bytes == 0) return DK_OK;
    hipPointerAttribute_t attr;
    const hipError_t h = hipPointerGetAttributes(&attr, user);
    if (h == hipSuccess && attr.type == hipMemoryTypeDevice) {
        b->dev = user;
        return DK_OK;
    }
    (void)hipGetLastError();                    // plain host memory is reported as an error: not one
    b->staged = true;
    return pool_alloc(e, bytes, (void **)&b->dev);
}

dk_status dk_reads_kmers(dk_engine *e, const dk_reads *r, uint64_t *kmers_lo, uint64_t *kmers_hi, uint64_t *hashes,
                         uint64_t *not_kmer, dk_stats *stats)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, r != nullptr && r->e == e);
    DK_TRY(reads_ready(e, r));
    const bool wide = e->cfg.k > 32;
    CHECK_ARG(e, r->n_bases == 0 || kmers_lo != nullptr);
    constexpr int KT = 256;                                  // 4 waves, 35 KiB of LDS: four workgroups per CU
    const uint64_t tile = (uint64_t)KT * (wide ? 8 : 16);
    const uint64_t n_tiles = (r->n_bases + tile - 1) / tile;
    if (n_tiles > 0xFFFFFFFFULL) return fail(e, DK_ERR_UNSUPPORTED, "batch too large for dk_reads_kmers");
    DK_HIP(e, hipSetDevice(e->device));
    DK_HIP(e, hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream));
    stage_begin(e);
    const size_t nb = r->n_bases * 8, nm = (r->n_bases + 63) / 64 * 8;
    OutBuf lo, hi, hs, nk;
    dk_status st = out_prepare(e, kmers_lo, nb, &lo);
    if (st == DK_OK && wide) st = out_prepare(e, kmers_hi, nb, &hi);
    if (st == DK_OK) st = out_prepare(e, hashes, nb, &hs);
    if (st == DK_OK) st = out_prepare(e, not_kmer, nm, &nk);
    hipError_t h = hipSuccess;
    if (st == DK_OK && r->n_bases) {
        const StreamView sv = view_of(r);
        const unsigned grid = (unsigned)std::min<uint64_t>(n_tiles, (uint64_t)e->n_cu * 4);
        if (wide && !e->opt.kmers_plain)
            kmers_tile_kernel<KT, true, true><<<grid, KT, 0, e->stream>>>(sv, (int)e->cfg.k, (int)e->cfg.canonical, e->cfg.seed, lo.dev,
                                                                          hi.dev, hs.dev, nk.dev, (uint32_t)n_tiles, e->d_ctr);
        else if (wide)
            kmers_tile_kernel<KT, true><<<grid, KT, 0, e->stream>>>(sv, (int)e->cfg.k, (int)e->cfg.canonical, e->cfg.seed, lo.dev,
                                                                    hi.dev, hs.dev, nk.dev, (uint32_t)n_tiles, e->d_ctr);
        else if (!e->opt.kmers_plain)
            kmers_tile_kernel<KT, false, true><<<grid, KT, 0, e->stream>>>(sv, (int)e->cfg.k, (int)e->cfg.canonical, e->cfg.seed, lo.dev,
                                                                           nullptr, hs.dev, nk.dev, (uint32_t)n_tiles, e->d_ctr);
        else
            kmers_tile_kernel<KT, false><<<grid, KT, 0, e->stream>>>(sv, (int)e->cfg.k, (int)e->cfg.canonical, e->cfg.seed, lo.dev,
                                                                     nullptr, hs.dev, nk.dev, (uint32_t)n_tiles, e->d_ctr);
        h = hipGetLastError();
        if (h == hipSuccess) stage_mark(e, "kmers");
        const OutBuf *bufs[4] = {&lo, &hi, &hs, &nk};
        const size_t sizes[4] = {nb, nb, nb, nm};
        for (int i = 0; i < 4 && h == hipSuccess; i++)
            if (bufs[i]->staged) h = hipMemcpyAsync(bufs[i]->user, bufs[i]->dev, sizes[i], hipMemcpyDeviceToHost, e->stream);
    }
    if (st == DK_OK && h == hipSuccess) st = read_counters(e);
    if (st == DK_OK && h == hipSuccess) st = stage_end(e);
    for (OutBuf *b : {&lo, &hi, &hs, &nk})
        if (b->staged) pool_free(e, b->dev);
    if (st != DK_OK) return st;
    if (h != hipSuccess) return fail(e, DK_ERR_HIP, "dk_reads_kmers failed: %s", hipGetErrorString(h));
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_reads = r->n_reads;
        stats->n_bases = r->n_bases;
        stats->n_windows = r->n_windows;
        stats->n_valid = e->h_ctr->n_valid;
    }
    return DK_OK;
}

void dk_reads_destroy(dk_reads *r)
{
    if (!r) return;
    if (r->ready) {
        (void)hipEventSynchronize(r->ready);       // the blocks go back to the pool: the upload must have left them
        (void)hipEventDestroy(r->ready);
    }
    if (r->owns) {
        pool_free(r->e, r->d_bases);
        pool_free(r->e, r->d_mask);
    }
    pool_free(r->e, r->d_uniform);
    delete r;
}

uint64_t dk_pack_ascii_host(const uint8_t *seq, const uint64_t *offsets, uint64_t n_reads,
                            uint64_t *bases, uint64_t *mask)
{
    uint64_t p = 0;
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t len = offsets[r + 1] - offsets[r];
        for (uint64_t j = 0; j <= len; j++, p++) {
            if ((p & 31) == 0) bases[p >> 5] = 0;
            if ((p & 63) == 0) mask[p >> 6] = 0;
            bool flag = true;
            uint64_t code = 0;
            if (j < len) {
                const uint32_t c = seq[offsets[r] + j] & 0xDFu;
                flag = !(c == 'A' || c == 'C' || c == 'G' || c == 'T');
                code = ((c >> 1) ^ (c >> 2)) & 3u;
            }
            if (flag) mask[p >> 6] |= 1ULL << (63 - (p & 63));
            else bases[p >> 5] |= code << (62 - 2 * (p & 31));
        }
    }
    return p;
}

static dk_status set_full(dk_engine *e)
{
    return fail(e, DK_ERR_SET_FULL,
                "exact set: %llu k-mer occurrences found no free slot in their 64-KiB segment; raise filter_log2_bits "
                "(the set now holds a subset of the k-mers given)", (unsigned long long)e->h_ctr->n_set_full);
}

// ---- KmerSet -------------------------------------------------------------------------------------
dk_status dk_set_create(dk_engine *e, dk_set **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr);
    *out = nullptr;
    DK_HIP(e, hipSetDevice(e->device));
    dk_set *s = new (std::nothrow) dk_set();
    if (!s) return fail(e, DK_ERR_OOM, "host allocation failed");
    s->e = e;
    s->n_bytes = (1ULL << e->cfg.filter_log2_bits) / 8;
    s->owns = true;
    s->d_words = nullptr;
    s->exact = e->cfg.set_kind == DK_SET_EXACT;
    dk_status st = pool_alloc(e, s->n_bytes, (void **)&s->d_words);
    if (st != DK_OK) { delete s; return st; }
    st = dk_set_clear(s);
    if (st != DK_OK) { dk_set_destroy(s); return st; }
    *out = s;
    return DK_OK;
}

dk_status dk_set_attach(dk_engine *e, void *d_filter, dk_set **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr && d_filter != nullptr);
    CHECK_ARG(e, ((uintptr_t)d_filter & 15) == 0);
    dk_set *s = new (std::nothrow) dk_set();
    if (!s) return fail(e, DK_ERR_OOM, "host allocation failed");
    s->e = e;
    s->n_bytes = (1ULL << e->cfg.filter_log2_bits) / 8;
    s->owns = false;
    s->exact = e->cfg.set_kind == DK_SET_EXACT;
    s->d_words = (unsigned long long *)d_filter;
    *out = s;
    return DK_OK;
}

dk_status dk_set_clear(dk_set *s)
{
    if (!s) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    DK_HIP(e, hipSetDevice(e->device));
    if (s->exact) {
        // an empty exact set is not all-zero: every slot holds its segment's EMPTY value
        const uint64_t n_words = s->n_bytes / 8;
        const int grid = grid_for(e, n_words / 2, DIRECT_BLOCK);
        if (e->cfg.k > 32) exact_clear_kernel<true><<<grid, DIRECT_BLOCK, 0, e->stream>>>(s->d_words, n_words, exact_T_of(e));
        else exact_clear_kernel<false><<<grid, DIRECT_BLOCK, 0, e->stream>>>(s->d_words, n_words, exact_T_of(e));
        DK_HIP(e, hipGetLastError());
    } else {
        DK_HIP(e, hipMemsetAsync(s->d_words, 0, s->n_bytes, e->stream));
    }
    DK_HIP(e, hipStreamSynchronize(e->stream));
    return DK_OK;
}

// the kernel family in force: the engine's configured mode unless option "mode" overrides it (1 = direct, 2 = bucketed;
// bench.py checks a small sample against the oracle through the family that a full-size batch takes)
static uint32_t mode_of(const dk_engine *e)
{
    return e->opt.mode == 1 ? (uint32_t)DK_MODE_DIRECT : e->opt.mode == 2 ? (uint32_t)DK_MODE_BUCKETED : e->cfg.mode;
}

static bool use_bucketed(const dk_engine *e, const dk_reads *r)
{
    if (mode_of(e) == DK_MODE_DIRECT) return false;
    if (mode_of(e) == DK_MODE_BUCKETED) return true;
    return dk::bucketed_pays(e, r->n_bases);
}

dk_status dk_set_insert(dk_set *s, const dk_reads *r, dk_stats *stats)
{
    if (!s || !r) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    CHECK_ARG(e, r->e == e);
    DK_HIP(e, hipSetDevice(e->device));
    DK_TRY(reads_ready(e, r));
    DK_HIP(e, hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream));
    stage_begin(e);
    if (r->n_bases) {
        bool direct = !use_bucketed(e, r);
        if (!direct) {
            const dk_status bs = dk::bucketed_insert(e, s, r);
            if (bs == DK_OK && e->h_ctr->n_set_full) return set_full(e);
            if (bs == DK_ERR_OVERFLOW) {
                // a bin overflowed (heavy-hitter k-mers): redo the batch with the direct family;
                // records already ORed in are harmless
                DK_HIP(e, hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream));
                stage_mark(e, "overflow_redo");
                direct = true;
            } else if (bs != DK_OK) {
                return bs;
            }
        }
        if (direct) {
            const StreamView sv = view_of(r);
            const FilterView fv = fview_of(e, s);
            const int grid = grid_for(e, r->n_bases, DIRECT_BLOCK);
            if (e->cfg.k > 32)
                insert_direct_kernel<true><<<grid, DIRECT_BLOCK, 0, e->stream>>>(sv, fv, (int)e->cfg.k, (int)e->cfg.canonical, e->d_ctr);
            else
                insert_direct_kernel<false><<<grid, DIRECT_BLOCK, 0, e->stream>>>(sv, fv, (int)e->cfg.k, (int)e->cfg.canonical, e->d_ctr);
            DK_HIP(e, hipGetLastError());
            stage_mark(e, "insert_direct");
        }
    }
    DK_TRY(read_counters(e));
    DK_TRY(stage_end(e));
    if (e->h_ctr->n_set_full) return set_full(e);
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_reads = r->n_reads;
        stats->n_bases = r->n_bases;
        stats->n_windows = r->n_windows;
        stats->n_valid = e->h_ctr->n_valid;
    }
    return DK_OK;
}

dk_status dk_set_contains(dk_set *s, const uint64_t *kmers_lo, const uint64_t *kmers_hi, uint64_t n, uint8_t *out)
{
    if (!s) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    const bool wide = e->cfg.k > 32;
    CHECK_ARG(e, n == 0 || (kmers_lo && out));
    CHECK_ARG(e, !wide || n == 0 || kmers_hi);
    if (n == 0) return DK_OK;
    DK_HIP(e, hipSetDevice(e->device));
    uint64_t *d_lo = nullptr, *d_hi = nullptr;
    uint8_t *d_out = nullptr;
    dk_status st = pool_alloc(e, n * 8, (void **)&d_lo);
    if (st == DK_OK && wide) st = pool_alloc(e, n * 8, (void **)&d_hi);
    if (st == DK_OK) st = pool_alloc(e, n, (void **)&d_out);
    hipError_t h = hipSuccess;
    if (st == DK_OK) {
        h = hipMemcpyAsync(d_lo, kmers_lo, n * 8, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess && wide) h = hipMemcpyAsync(d_hi, kmers_hi, n * 8, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess) {
            const FilterView fv = fview_of(e, s);
            const uint64_t grid = (n + DIRECT_BLOCK - 1) / DIRECT_BLOCK;
            if (wide) contains_kernel<true><<<(int)grid, DIRECT_BLOCK, 0, e->stream>>>(fv, d_lo, d_hi, n, d_out);
            else contains_kernel<false><<<(int)grid, DIRECT_BLOCK, 0, e->stream>>>(fv, d_lo, d_hi, n, d_out);
            h = hipGetLastError();
        }
        if (h == hipSuccess) h = hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, e->stream);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
    }
    pool_free(e, d_lo);
    pool_free(e, d_hi);
    pool_free(e, d_out);
    if (st != DK_OK) return st;
    if (h != hipSuccess) return fail(e, DK_ERR_HIP, "dk_set_contains failed: %s", hipGetErrorString(h));
    return DK_OK;
}

dk_status dk_set_device_ptr(dk_set *s, void **d_filter, uint64_t *n_bytes)
{
    if (!s) return DK_ERR_INVALID_ARG;
    if (d_filter) *d_filter = s->d_words;
    if (n_bytes) *n_bytes = s->n_bytes;
    return DK_OK;
}

dk_status dk_set_download(dk_set *s, uint64_t *words)
{
    if (!s || !words) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    DK_HIP(e, hipSetDevice(e->device));
    DK_HIP(e, hipMemcpyAsync(words, s->d_words, s->n_bytes, hipMemcpyDeviceToHost, e->stream));
    DK_HIP(e, hipStreamSynchronize(e->stream));
    return DK_OK;
}

dk_status dk_set_upload(dk_set *s, const uint64_t *words)
{
    if (!s || !words) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    DK_HIP(e, hipSetDevice(e->device));
    DK_HIP(e, hipMemcpyAsync(s->d_words, words, s->n_bytes, hipMemcpyHostToDevice, e->stream));
    DK_HIP(e, hipStreamSynchronize(e->stream));
    return DK_OK;
}

dk_status dk_set_popcount(dk_set *s, uint64_t *n_bits_set)
{
    if (!s || !n_bits_set) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    DK_HIP(e, hipSetDevice(e->device));
    DK_HIP(e, hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream));
    const uint64_t n_words = s->n_bytes / 8;
    if (s->exact) {
        const int grid = grid_for(e, n_words / 2, DIRECT_BLOCK);
        if (e->cfg.k > 32)
            exact_count_kernel<true><<<grid, DIRECT_BLOCK, 0, e->stream>>>(s->d_words, n_words, exact_T_of(e), &e->d_ctr->n_valid);
        else
            exact_count_kernel<false><<<grid, DIRECT_BLOCK, 0, e->stream>>>(s->d_words, n_words, exact_T_of(e), &e->d_ctr->n_valid);
    } else {
        popcount_kernel<<<grid_for(e, n_words, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
            (const uint64_t *)s->d_words, n_words, &e->d_ctr->n_valid);
    }
    DK_HIP(e, hipGetLastError());
    DK_TRY(read_counters(e));
    *n_bits_set = e->h_ctr->n_valid;
    return DK_OK;
}

struct dk_filter_header {
    char magic[8];                 // "DKBLOOM1" | "DKEXACT1"
    uint32_t k, canonical, filter_log2_bits, n_hashes;
    uint64_t seed;
    uint64_t n_bytes;
    uint8_t reserved[24];
};
static_assert(sizeof(dk_filter_header) == 64, "filter file header is 64 bytes");

static dk_filter_header header_of(const dk_set *s)
{
    dk_filter_header h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, s->exact ? "DKEXACT1" : "DKBLOOM1", 8);
    h.k = s->e->cfg.k;
    h.canonical = s->e->cfg.canonical;
    h.filter_log2_bits = s->e->cfg.filter_log2_bits;
    h.n_hashes = s->exact ? 0 : s->e->cfg.n_hashes;
    h.seed = s->e->cfg.seed;
    h.n_bytes = s->n_bytes;
    return h;
}

dk_status dk_set_save(dk_set *s, const char *path)
{
    if (!s || !path) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    DK_HIP(e, hipSetDevice(e->device));
    FILE *f = fopen(path, "wb");
    if (!f) return fail(e, DK_ERR_INVALID_ARG, "cannot open %s for writing", path);
    const dk_filter_header hd = header_of(s);
    const size_t chunk = std::min<uint64_t>(s->n_bytes, 64ULL << 20);
    void *buf = nullptr;
    dk_status st = DK_OK;
    if (hipHostMalloc(&buf, chunk, hipHostMallocDefault) != hipSuccess) st = fail(e, DK_ERR_OOM, "staging buffer allocation failed");
    if (st == DK_OK && fwrite(&hd, sizeof hd, 1, f) != 1) st = fail(e, DK_ERR_INVALID_ARG, "write to %s failed", path);
    for (uint64_t off = 0; st == DK_OK && off < s->n_bytes; off += chunk) {
        const size_t nb = (size_t)std::min<uint64_t>(chunk, s->n_bytes - off);
        hipError_t h = hipMemcpyAsync(buf, (const char *)s->d_words + off, nb, hipMemcpyDeviceToHost, e->stream);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "reading the filter back failed: %s", hipGetErrorString(h));
        else if (fwrite(buf, 1, nb, f) != nb) st = fail(e, DK_ERR_INVALID_ARG, "write to %s failed", path);
    }
    if (buf) (void)hipHostFree(buf);
    if (fclose(f) != 0 && st == DK_OK) st = fail(e, DK_ERR_INVALID_ARG, "closing %s failed", path);
    return st;
}

dk_status dk_set_load(dk_set *s, const char *path)
{
    if (!s || !path) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    DK_HIP(e, hipSetDevice(e->device));
    FILE *f = fopen(path, "rb");
    if (!f) return fail(e, DK_ERR_INVALID_ARG, "cannot open %s", path);
    dk_filter_header hd;
    const dk_filter_header want = header_of(s);
    dk_status st = DK_OK;
    if (fread(&hd, sizeof hd, 1, f) != 1 || memcmp(hd.magic, want.magic, 8) != 0)
        st = fail(e, DK_ERR_INVALID_ARG, "%s is not a %s file", path, s->exact ? "DKEXACT1 exact-set" : "DKBLOOM1 filter");
    else if (hd.k != want.k || hd.canonical != want.canonical || hd.filter_log2_bits != want.filter_log2_bits ||
             hd.n_hashes != want.n_hashes || hd.seed != want.seed || hd.n_bytes != want.n_bytes)
        st = fail(e, DK_ERR_INVALID_ARG, "%s was built with another geometry (k=%u log2_bits=%u n_hashes=%u)", path, hd.k,
                  hd.filter_log2_bits, hd.n_hashes);
    const size_t chunk = std::min<uint64_t>(s->n_bytes, 64ULL << 20);
    void *buf = nullptr;
    if (st == DK_OK && hipHostMalloc(&buf, chunk, hipHostMallocDefault) != hipSuccess)
        st = fail(e, DK_ERR_OOM, "staging buffer allocation failed");
    for (uint64_t off = 0; st == DK_OK && off < s->n_bytes; off += chunk) {
        const size_t nb = (size_t)std::min<uint64_t>(chunk, s->n_bytes - off);
        if (fread(buf, 1, nb, f) != nb) { st = fail(e, DK_ERR_INVALID_ARG, "%s is truncated", path); break; }
        hipError_t h = hipMemcpyAsync((char *)s->d_words + off, buf, nb, hipMemcpyHostToDevice, e->stream);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "uploading the filter failed: %s", hipGetErrorString(h));
    }
    if (buf) (void)hipHostFree(buf);
    fclose(f);
    return st;
}

dk_status dk_or_reduce_slices(dk_engine *e, void *d_dst, const void *d_src, uint64_t n_slices, uint64_t slice_bytes)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, d_dst != nullptr && (d_src != nullptr || n_slices == 0));
    CHECK_ARG(e, slice_bytes % 16 == 0);
    CHECK_ARG(e, ((uintptr_t)d_dst & 15) == 0 && ((uintptr_t)d_src & 15) == 0);
    if (n_slices == 0 || slice_bytes == 0) return DK_OK;
    DK_HIP(e, hipSetDevice(e->device));
    const uint64_t vec = slice_bytes / 16;
    or_slices_kernel<<<grid_for(e, vec, DIRECT_BLOCK), DIRECT_BLOCK, 0, e->stream>>>(
        (uint4 *)d_dst, (const uint4 *)d_src, n_slices, vec);
    DK_HIP(e, hipGetLastError());
    DK_HIP(e, hipStreamSynchronize(e->stream));
    return DK_OK;
}

dk_status dk_union_slices(dk_engine *e, void *d_dst, const void *d_src, uint64_t n_slices, uint64_t slice_bytes,
                          uint64_t first_segment)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, e->cfg.set_kind == DK_SET_EXACT);
    CHECK_ARG(e, d_dst != nullptr && (d_src != nullptr || n_slices == 0));
    CHECK_ARG(e, slice_bytes % SEG_BYTES == 0);
    CHECK_ARG(e, ((uintptr_t)d_dst & 15) == 0 && ((uintptr_t)d_src & 15) == 0);
    const int T = exact_T_of(e);
    const uint64_t n_seg = slice_bytes / SEG_BYTES;
    CHECK_ARG(e, first_segment + n_seg <= (1ULL << T) && n_seg <= 0x7FFFFFFFULL);
    if (n_slices == 0 || n_seg == 0) return DK_OK;
    DK_HIP(e, hipSetDevice(e->device));
    DK_HIP(e, hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream));
    if (e->cfg.k > 32)
        union_slices_kernel<true><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(
            (unsigned long long *)d_dst, (const unsigned long long *)d_src, n_slices, slice_bytes / 8, first_segment, T, e->d_ctr);
    else
        union_slices_kernel<false><<<(unsigned)n_seg, SEG_THREADS, 0, e->stream>>>(
            (unsigned long long *)d_dst, (const unsigned long long *)d_src, n_slices, slice_bytes / 8, first_segment, T, e->d_ctr);
    DK_HIP(e, hipGetLastError());
    DK_TRY(read_counters(e));
    if (e->h_ctr->n_set_full) return set_full(e);
    return DK_OK;
}

// ---- multi-GPU -----------------------------------------------------------------------------------------------
dk_status dk_comm_unique_id(uint8_t *id)
{
    if (!id) return fail(nullptr, DK_ERR_INVALID_ARG, "id is NULL");
    RcclApi *api = rccl();
    if (!api->lib) return fail(nullptr, DK_ERR_UNSUPPORTED, "%s", api->err.c_str());
    ncclUniqueId_t u;
    const int r = api->GetUniqueId(&u);
    if (r != RCCL_SUCCESS) return fail(nullptr, DK_ERR_HIP, "ncclGetUniqueId failed: %s", api->GetErrorString(r));
    static_assert(sizeof u == DK_COMM_ID_BYTES, "ncclUniqueId is 128 bytes (rccl.h:40)");
    memcpy(id, &u, sizeof u);
    return DK_OK;
}

dk_status dk_comm_init(dk_engine *e, const uint8_t *id, uint32_t rank, uint32_t world_size)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, world_size >= 1 && rank < world_size);
    CHECK_ARG(e, id != nullptr || world_size == 1);
    CHECK_ARG(e, e->comm == nullptr);
    DK_HIP(e, hipSetDevice(e->device));
    dk_comm *c = new (std::nothrow) dk_comm();
    if (!c) return fail(e, DK_ERR_OOM, "host allocation failed");
    c->comm = nullptr;
    c->rank = rank;
    c->world = world_size;
    c->staging = nullptr;
    c->staging_bytes = 0;
    c->dead = false;
    if (world_size > 1 || id != nullptr) {       // (one rank WITH an id: a real communicator of one, see set_allreduce)
        RcclApi *api = rccl();
        if (!api->lib) { delete c; return fail(e, DK_ERR_UNSUPPORTED, "%s", api->err.c_str()); }
        ncclUniqueId_t u;
        memcpy(&u, id, sizeof u);
        const int r = api->CommInitRank(&c->comm, (int)world_size, u, (int)rank);
        if (r != RCCL_SUCCESS) { delete c; return fail(e, DK_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString(r)); }
        c->staging_bytes = e->opt.comm_staging_kb ? (uint64_t)e->opt.comm_staging_kb << 10 : 1ULL << 30;
        const dk_status st = pool_alloc(e, c->staging_bytes, &c->staging);
        if (st != DK_OK) { (void)api->CommDestroy(c->comm); delete c; return st; }
    }
    e->comm = c;
    e->cfg.rank = rank;
    e->cfg.world_size = world_size;
    return DK_OK;
}

dk_status dk_comm_finalize(dk_engine *e)
{
    if (!e) return DK_ERR_INVALID_ARG;
    if (!e->comm) return DK_OK;
    if (e->comm->comm) (void)rccl()->CommDestroy(e->comm->comm);
    pool_free(e, e->comm->staging);
    delete e->comm;
    e->comm = nullptr;
    return DK_OK;
}

dk_status dk_set_allreduce_or(dk_set *s, uint64_t *bytes_sent)
{
    if (!s) return DK_ERR_INVALID_ARG;
    dk_engine *e = s->e;
    DK_HIP(e, hipSetDevice(e->device));
    DK_TRY(dk::set_allreduce(e, s, bytes_sent));
    if (e->comm && e->comm->comm && e->h_ctr->n_set_full) return set_full(e);
    return DK_OK;
}

void dk_set_destroy(dk_set *s)
{
    if (!s) return;
    if (s->owns) pool_free(s->e, s->d_words);
    delete s;
}

// ---- membership pass / KmerCounter ----------------------------------------------------------------
dk_status dk_probe(dk_engine *e, dk_set *s, const dk_reads *r, dk_result **out, dk_stats *stats)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, r != nullptr && out != nullptr);
    CHECK_ARG(e, r->e == e && (!s || s->e == e));
    *out = nullptr;
    DK_HIP(e, hipSetDevice(e->device));
    DK_TRY(reads_ready(e, r));
    dk_result *res = new (std::nothrow) dk_result();
    if (!res) return fail(e, DK_ERR_OOM, "host allocation failed");
    res->e = e;
    res->d_lo = res->d_hi = nullptr;
    res->d_cnt = nullptr;
    res->n = 0;
    res->n_regions = 1;
    res->region_cap = 0;
    memset(res->region_n, 0, sizeof res->region_n);
    res->wide = e->cfg.k > 32;
    res->owns = true;
    hipError_t h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
    if (h != hipSuccess) { delete res; return fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h)); }
    memset(e->h_ctr, 0, sizeof(Counters));
    stage_begin(e);
    dk_status st = DK_OK;
    if (r->n_bases) {
        // KmerCounter (no set) partitions by batch size, not by the filter geometry: the bucketed family
        // pays from a few million positions on, whatever the engine's filter size
        bool direct = s ? !use_bucketed(e, r)
                        : mode_of(e) == DK_MODE_DIRECT || (mode_of(e) == DK_MODE_AUTO && r->n_bases < (1ULL << 22));
        if (!direct) {
            st = dk::bucketed_probe(e, s, r, res);
            if (st == DK_ERR_OVERFLOW) {
                // exactness first: drop the partial result and redo the batch with the direct family
                pool_free(e, res->d_lo);
                pool_free(e, res->d_hi);
                pool_free(e, res->d_cnt);
                res->d_lo = nullptr;
                res->d_hi = nullptr;
                res->d_cnt = nullptr;
                res->n = 0;
                res->n_regions = 1;
                res->region_cap = 0;
                h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
                st = h == hipSuccess ? DK_OK : fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h));
                stage_mark(e, "overflow_redo");
                direct = true;
            }
        }
        if (direct && st == DK_OK) st = res->wide ? probe_direct<true>(e, s, r, res) : probe_direct<false>(e, s, r, res);
    }
    if (st == DK_OK) st = stage_end(e);
    if (st != DK_OK) { dk_result_destroy(res); return st; }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_reads = r->n_reads;
        stats->n_bases = r->n_bases;
        stats->n_windows = r->n_windows;
        stats->n_valid = e->h_ctr->n_valid;
        stats->n_absent = e->h_ctr->n_absent;
        stats->n_distinct = e->h_ctr->n_distinct;
        stats->n_emitted = res->n;
    }
    *out = res;
    return DK_OK;
}

// ---- child-only accumulator ---------------------------------------------------------------------------------
static dk_result *result_new(dk_engine *e)
{
    dk_result *res = new (std::nothrow) dk_result();
    if (!res) return nullptr;
    res->e = e;
    res->d_lo = res->d_hi = nullptr;
    res->d_cnt = nullptr;
    res->n = 0;
    res->n_regions = 1;
    res->region_cap = 0;
    memset(res->region_n, 0, sizeof res->region_n);
    res->wide = e->cfg.k > 32;
    res->owns = true;
    return res;
}

static dk_status accum_clear(dk_accum *a)
{
    dk_engine *e = a->e;
    DK_HIP(e, hipMemsetAsync(a->fill, 0, a->n_units * 4, e->stream));
    DK_HIP(e, hipMemsetAsync(a->d_novf, 0, 8, e->stream));
    DK_HIP(e, hipStreamSynchronize(e->stream));
    a->n_absent = a->n_valid = a->n_windows = a->n_reads = a->n_bases = a->n_batches = 0;
    a->failed = false;
    a->exchanged = false;
    return DK_OK;
}

dk_status dk_accum_create(dk_engine *e, dk_set *s, uint32_t window_index, uint32_t window_count,
                          uint64_t capacity_records, dk_accum **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr);
    *out = nullptr;
    CHECK_ARG(e, !s || s->e == e);
    CHECK_ARG(e, window_count >= 1 && (window_count & (window_count - 1)) == 0);
    CHECK_ARG(e, window_index < window_count);
    CHECK_ARG(e, capacity_records >= 1);
    const int T = set_segment_bits(e);
    if (T < 1 || T > MAX_SEG_BITS)
        return fail(e, DK_ERR_UNSUPPORTED, "accumulators need a set geometry of 2^20..2^%d bits", MAX_SEG_BITS + 19);
    int wbits = 0;
    while ((1u << wbits) < window_count) wbits++;
    if (wbits > T - 1)
        return fail(e, DK_ERR_INVALID_ARG, "window_count %u: at most %u windows with this set geometry (each covers at least two 64-KiB segments)",
                    window_count, 1u << (T - 1));
    const bool wide = e->cfg.k > 32;
    // counting units: 2^u per segment, so that a unit's records fit the registers of one seg_count workgroup
    const uint64_t n_seg_w = 1ULL << (T - wbits);
    const uint64_t unit_target = wide ? 6144 : 12288;
    int u = std::min(e->opt.accum_min_u, MAX_SUB_BITS);
    while (u < MAX_SUB_BITS && capacity_records / (n_seg_w << u) > unit_target) u++;
    if (capacity_records / (n_seg_w << u) > unit_target)
        return fail(e, DK_ERR_UNSUPPORTED, "capacity %llu is more than %llu records per hash window of this set geometry: use more windows",
                    (unsigned long long)capacity_records, (unsigned long long)(unit_target * (n_seg_w << MAX_SUB_BITS)));
    DK_HIP(e, hipSetDevice(e->device));
    dk_accum *a = new (std::nothrow) dk_accum();
    if (!a) return fail(e, DK_ERR_OOM, "host allocation failed");
    a->e = e;
    a->s = s;
    a->wbits = wbits;
    a->widx = window_index;
    a->T = T;
    a->u = u;
    a->n_units = n_seg_w << u;
    a->wide = wide;
    // packed unit records (6 bytes): whenever the unit's hash prefix covers at least 16 bits, i.e. the remaining 48 fit
    a->packed = !wide && !e->opt.accum_plain && T + u >= PACKED_MIN_PREFIX_BITS;
    // room per unit = mean + 5 sigma of a Poisson fill: absent occurrences are sequencing errors, i.e. independent;
    // a unit that still runs full sends the rest to the overflow list, which is counted with it (exact either way).
    // Without a set (KmerCounter over batches) every copy of a k-mer lands in one unit: the variance is then the mean times
    // the multiplicity (piece_capacity's ratio), so sigma is scaled by the same segment_ratio the partition uses.
    {
        const double mean = (double)capacity_records / (double)a->n_units;
        // unit stride = a multiple of 4 KiB plus 128 bytes, i.e. never a power of two: the count kernel's workgroups read
        // consecutive units at the same time, and with a stride of exactly 128 KiB all of those reads land on the same HBM
        // channels -- 2966 ms instead of 55 for the 9.6 x 10^9 records of a whole-genome pass; every other stride tried
        // (20..32 x 4 KiB + 128 B, 96 KiB exactly, odd sizes) takes the same 55 ms (tools/experiments/accum_count.py).
        // The smallest such capacity that leaves 5 sigma of room (the rare unit beyond it spills to the overflow list).
        // Packed units (6-byte records): whole 64-record blocks, and a stride in bytes that is not a multiple of 16 KiB.
        const double sig = s ? 1.0 : sqrt(segment_ratio(e));
        const uint32_t need = (uint32_t)(mean + 5.0 * sig * sqrt(mean + 1.0)) + 1u;
        if (a->packed) {
            a->unit_cap = (need + 63) / 64 * 64;
            if (((uint64_t)a->unit_cap * PACKED_REC_BYTES) % 16384 == 0) a->unit_cap += 64;
        } else {
            const uint32_t rec_bytes = wide ? 16u : 8u, per_4k = 4096u / rec_bytes, odd = 128u / rec_bytes;
            a->unit_cap = (need > odd ? (need - odd + per_4k - 1) / per_4k * per_4k : 0u) + odd;
        }
        if ((uint32_t)e->opt.accum_unit_cap >= need) a->unit_cap = (uint32_t)e->opt.accum_unit_cap;
        if (a->packed) a->unit_cap = (a->unit_cap + 63) / 64 * 64;      // (a forced capacity too: whole 64-record blocks)
    }
    a->store = nullptr;
    a->fill = nullptr;
    a->ovf = nullptr;
    a->d_novf = nullptr;
    a->ovf_cap = std::max<uint64_t>(1ULL << 16, capacity_records / 64);
    const size_t rb = accum_rec_bytes(a), ovf_rb = wide ? sizeof(Rec2) : sizeof(Rec1);    // (the overflow list holds plain records)
    dk_status st = pool_alloc(e, a->n_units * (uint64_t)a->unit_cap * rb, &a->store);
    if (st == DK_OK) st = pool_alloc(e, a->n_units * 4, (void **)&a->fill);
    if (st == DK_OK) st = pool_alloc(e, a->ovf_cap * ovf_rb, &a->ovf);
    if (st == DK_OK) st = pool_alloc(e, 256, (void **)&a->d_novf);
    if (st == DK_OK) st = accum_clear(a);
    if (st != DK_OK) { dk_accum_destroy(a); return st; }
    *out = a;
    return DK_OK;
}

void dk_accum_destroy(dk_accum *a)
{
    if (!a) return;
    pool_free(a->e, a->store);
    pool_free(a->e, a->fill);
    pool_free(a->e, a->ovf);
    pool_free(a->e, a->d_novf);
    delete a;
}

dk_status dk_accum_reset(dk_accum *a, uint32_t window_index)
{
    if (!a) return DK_ERR_INVALID_ARG;
    dk_engine *e = a->e;
    CHECK_ARG(e, window_index < (1u << a->wbits));
    DK_HIP(e, hipSetDevice(e->device));
    a->widx = window_index;
    return accum_clear(a);
}

dk_status dk_accum_stats(const dk_accum *a, dk_stats *out)
{
    if (!a || !out) return DK_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    out->n_reads = a->n_reads;
    out->n_bases = a->n_bases;
    out->n_windows = a->n_windows;
    out->n_valid = a->n_valid;
    out->n_absent = a->n_absent;
    return DK_OK;
}

dk_status dk_accum_device_bytes(const dk_accum *a, uint64_t *n_bytes)
{
    if (!a || !n_bytes) return DK_ERR_INVALID_ARG;
    const size_t rb = accum_rec_bytes(a), ovf_rb = a->wide ? sizeof(Rec2) : sizeof(Rec1);
    *n_bytes = a->n_units * (uint64_t)a->unit_cap * rb + a->n_units * 4 + a->ovf_cap * ovf_rb + 256;
    return DK_OK;
}

dk_status dk_accum_add(dk_accum *a, const dk_reads *r, dk_stats *stats)
{
    if (!a || !r) return DK_ERR_INVALID_ARG;
    dk_engine *e = a->e;
    CHECK_ARG(e, r->e == e);
    if (a->failed) return fail(e, DK_ERR_OVERFLOW, "the accumulator lost records in an earlier call: dk_accum_reset it first");
    if (a->exchanged) return fail(e, DK_ERR_INVALID_ARG, "the accumulator was consumed by dk_accum_exchange_finish: dk_accum_reset it first");
    DK_HIP(e, hipSetDevice(e->device));
    DK_TRY(reads_ready(e, r));
    DK_HIP(e, hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream));
    memset(e->h_ctr, 0, sizeof(Counters));
    stage_begin(e);
    dk_status st = DK_OK;
    if (r->n_bases) {
        bool direct = mode_of(e) == DK_MODE_DIRECT ||
                      (mode_of(e) == DK_MODE_AUTO && (a->s ? !dk::bucketed_pays(e, r->n_bases, a->wbits) : r->n_bases < (1ULL << 22)));
        uint64_t redo_from = 0, absent_done = 0;
        if (!direct) {
            bool fatal = false;
            st = e->cfg.k > 32 ? dk::bucketed_accum_add_t<true>(e, a, r, &fatal, &redo_from, &absent_done)
                               : dk::bucketed_accum_add_t<false>(e, a, r, &fatal, &redo_from, &absent_done);
            if (st == DK_ERR_OVERFLOW && !fatal) {
                // the partition lost records: what it had not completed (hashes >= redo_from) is redone exactly
                hipError_t h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
                st = h == hipSuccess ? DK_OK : fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h));
                memset(e->h_ctr, 0, sizeof(Counters));
                stage_mark(e, "overflow_redo");
                direct = true;
            } else if (st != DK_OK && fatal) {
                a->failed = true;
            }
        }
        if (direct && st == DK_OK) {
            st = e->cfg.k > 32 ? accum_add_direct<true>(e, a, r, redo_from) : accum_add_direct<false>(e, a, r, redo_from);
            if (st != DK_OK) a->failed = true;
            else e->h_ctr->n_absent += absent_done;
        }
    }
    if (st == DK_ERR_OVERFLOW && a->failed)
        st = fail(e, DK_ERR_OVERFLOW, "accumulator full: %llu occurrences found neither room in their unit nor in the overflow list "
                  "(capacity too small for this sample: use more windows or a larger capacity)",
                  (unsigned long long)(e->h_ctr->n_sink_drop + e->h_ctr->n_overflow));
    if (st == DK_OK) st = stage_end(e);
    if (st != DK_OK) return st;
    a->n_batches++;
    a->n_reads += r->n_reads;
    a->n_bases += r->n_bases;
    a->n_windows += r->n_windows;
    a->n_valid += e->h_ctr->n_valid;
    a->n_absent += e->h_ctr->n_absent;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_reads = r->n_reads;
        stats->n_bases = r->n_bases;
        stats->n_windows = r->n_windows;
        stats->n_valid = e->h_ctr->n_valid;
        stats->n_absent = e->h_ctr->n_absent;
    }
    return DK_OK;
}

dk_status dk_accum_finish(dk_accum *a, uint32_t min_count, dk_result **out, dk_stats *stats)
{
    if (!a) return DK_ERR_INVALID_ARG;
    dk_engine *e = a->e;
    CHECK_ARG(e, out != nullptr && min_count >= 1);
    *out = nullptr;
    if (a->failed) return fail(e, DK_ERR_OVERFLOW, "the accumulator lost records in an earlier call: dk_accum_reset it first");
    if (a->exchanged) return fail(e, DK_ERR_INVALID_ARG, "the accumulator was consumed by dk_accum_exchange_finish: dk_accum_reset it first");
    DK_HIP(e, hipSetDevice(e->device));
    dk_result *res = result_new(e);
    if (!res) return fail(e, DK_ERR_OOM, "host allocation failed");
    hipError_t h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
    if (h != hipSuccess) { delete res; return fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h)); }
    memset(e->h_ctr, 0, sizeof(Counters));
    stage_begin(e);
    dk_status st = a->wide ? accum_finish_t<true>(e, a, min_count, res, nullptr, nullptr, 0, 0, a->n_units, nullptr, 0, nullptr)
                           : accum_finish_t<false>(e, a, min_count, res, nullptr, nullptr, 0, 0, a->n_units, nullptr, 0, nullptr);
    if (st == DK_OK) st = stage_end(e);
    if (st != DK_OK) { dk_result_destroy(res); return st; }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_reads = a->n_reads;
        stats->n_bases = a->n_bases;
        stats->n_windows = a->n_windows;
        stats->n_valid = a->n_valid;
        stats->n_absent = a->n_absent;
        stats->n_distinct = e->h_ctr->n_distinct;
        stats->n_emitted = res->n;
    }
    *out = res;
    return DK_OK;
}

dk_status dk_accum_geometry(const dk_accum *a, uint64_t *n_units, uint32_t *unit_cap, uint32_t *record_bytes)
{
    if (!a) return DK_ERR_INVALID_ARG;
    if (n_units) *n_units = a->n_units;
    if (unit_cap) *unit_cap = a->unit_cap;
    if (record_bytes) *record_bytes = (uint32_t)accum_rec_bytes(a);
    return DK_OK;
}

dk_status dk_accum_device_view(dk_accum *a, void **d_store, void **d_fill, void **d_overflow, uint64_t *n_overflow)
{
    if (!a) return DK_ERR_INVALID_ARG;
    dk_engine *e = a->e;
    if (a->failed) return fail(e, DK_ERR_OVERFLOW, "the accumulator lost records in an earlier call: dk_accum_reset it first");
    if (a->exchanged) return fail(e, DK_ERR_INVALID_ARG, "the accumulator was consumed by dk_accum_exchange_finish: dk_accum_reset it first");
    DK_HIP(e, hipSetDevice(e->device));
    unsigned long long n = 0;
    DK_HIP(e, hipMemcpyAsync(&n, a->d_novf, 8, hipMemcpyDeviceToHost, e->stream));
    DK_HIP(e, hipStreamSynchronize(e->stream));           // also: every dk_accum_add has landed in the store
    if (d_store) *d_store = a->store;
    if (d_fill) *d_fill = a->fill;
    if (d_overflow) *d_overflow = a->ovf;
    if (n_overflow) *n_overflow = n > a->ovf_cap ? a->ovf_cap : n;
    return DK_OK;
}

dk_status dk_accum_finish_pieces(dk_accum *a, const void *d_stores, const void *d_fills, uint32_t n_pieces,
                                 uint64_t first_unit, uint64_t n_units, const void *d_extra, uint64_t n_extra,
                                 uint32_t min_count, dk_result **out, dk_stats *stats)
{
    if (!a) return DK_ERR_INVALID_ARG;
    dk_engine *e = a->e;
    CHECK_ARG(e, out != nullptr && min_count >= 1);
    *out = nullptr;
    CHECK_ARG(e, d_stores != nullptr && d_fills != nullptr);
    CHECK_ARG(e, n_pieces >= 1 && n_pieces <= (uint32_t)MAX_R);
    CHECK_ARG(e, n_units >= 1 && first_unit + n_units <= a->n_units);
    CHECK_ARG(e, d_extra != nullptr || n_extra == 0);
    DK_HIP(e, hipSetDevice(e->device));
    dk_result *res = result_new(e);
    if (!res) return fail(e, DK_ERR_OOM, "host allocation failed");
    hipError_t h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
    if (h != hipSuccess) { delete res; return fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h)); }
    memset(e->h_ctr, 0, sizeof(Counters));
    stage_begin(e);
    uint64_t n_records = 0;
    dk_status st = a->wide ? accum_finish_t<true>(e, a, min_count, res, d_stores, (const uint32_t *)d_fills, n_pieces, first_unit,
                                                  n_units, d_extra, n_extra, &n_records)
                           : accum_finish_t<false>(e, a, min_count, res, d_stores, (const uint32_t *)d_fills, n_pieces, first_unit,
                                                   n_units, d_extra, n_extra, &n_records);
    if (st == DK_OK) st = stage_end(e);
    if (st != DK_OK) { dk_result_destroy(res); return st; }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_absent = n_records;
        stats->n_distinct = e->h_ctr->n_distinct;
        stats->n_emitted = res->n;
    }
    *out = res;
    return DK_OK;
}

// Multi-GPU end of a child pass behind the C ABI: every rank has accumulated ITS child reads over
// the same hash window; the ranks swap unit ranges on the engine's communicator -- in place, piece by piece through the
// staging buffer, so no second copy of the store exists -- and each rank counts its own share of the units from the P
// pieces.  Collective: every rank of the communicator must call it.
dk_status dk_accum_exchange_finish(dk_accum *a, uint32_t min_count, dk_result **out, dk_stats *stats, uint64_t *bytes_sent)
{
    if (!a) return DK_ERR_INVALID_ARG;
    dk_engine *e = a->e;
    CHECK_ARG(e, out != nullptr && min_count >= 1);
    *out = nullptr;
    if (bytes_sent) *bytes_sent = 0;
    dk_comm *c = e->comm;
    if (c && c->dead) return fail(e, DK_ERR_HIP, "the communicator was aborted by an earlier failure");
    if (!c || !c->comm) {
        if (e->cfg.world_size > 1) return fail(e, DK_ERR_INVALID_ARG, "no communicator: dk_comm_init first");
        return dk_accum_finish(a, min_count, out, stats);
    }
    if (a->exchanged) return fail(e, DK_ERR_INVALID_ARG, "the accumulator was consumed by dk_accum_exchange_finish: dk_accum_reset it first");
    DK_HIP(e, hipSetDevice(e->device));
    RcclApi *api = rccl();
    const uint64_t P = c->world, r = c->rank;
    CHECK_ARG(e, P <= (uint64_t)MAX_R);
    if (a->n_units % P != 0) return fail(e, DK_ERR_UNSUPPORTED, "%llu counting units do not split over %llu ranks",
                                         (unsigned long long)a->n_units, (unsigned long long)P);
    const size_t rb = accum_rec_bytes(a), ob = a->wide ? sizeof(Rec2) : sizeof(Rec1);
    // 1. one 64-byte header per rank, all-gathered: the ranks must have built their accumulators alike, and a rank whose
    //    accumulator failed makes every rank return an error here instead of leaving the others inside a collective
    unsigned long long my_ovf = 0;
    DK_HIP(e, hipMemcpyAsync(&my_ovf, a->d_novf, 8, hipMemcpyDeviceToHost, e->stream));
    DK_HIP(e, hipStreamSynchronize(e->stream));
    if (my_ovf > a->ovf_cap) my_ovf = a->ovf_cap;
    constexpr int HW = 8;
    uint64_t hdr[MAX_R * HW] = {0};
    uint64_t *mine = hdr + r * HW;
    mine[0] = a->n_units; mine[1] = a->unit_cap; mine[2] = rb; mine[3] = ((uint64_t)a->wbits << 32) | a->widx;
    mine[4] = a->failed ? 1 : 0; mine[5] = my_ovf;
    uint64_t *d_hdr = nullptr;
    DK_TRY(pool_alloc(e, sizeof hdr, (void **)&d_hdr));
    dk_status st = DK_OK;
    hipError_t h = hipMemcpyAsync(d_hdr + r * HW, mine, HW * 8, hipMemcpyHostToDevice, e->stream);
    if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "header upload failed: %s", hipGetErrorString(h));
    if (st == DK_OK) {
        const int rr = api->AllGather(d_hdr + r * HW, d_hdr, HW * 8, RCCL_UINT8, c->comm, e->stream);
        if (rr != RCCL_SUCCESS) st = comm_fail(e, false, "ncclAllGather(header)", rr, __FILE__, __LINE__);
    }
    if (st == DK_OK) {
        h = hipMemcpyAsync(hdr, d_hdr, P * HW * 8, hipMemcpyDeviceToHost, e->stream);
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "header exchange failed: %s", hipGetErrorString(h));
    }
    pool_free(e, d_hdr);
    if (st != DK_OK) return st;
    uint64_t ovf_max = 0, ovf_sum = 0;
    for (uint64_t q = 0; q < P; q++) {
        const uint64_t *hq = hdr + q * HW;
        if (hq[4]) return fail(e, DK_ERR_OVERFLOW, "rank %llu's accumulator lost records in an earlier call: every rank stops here",
                               (unsigned long long)q);
        if (hq[0] != mine[0] || hq[1] != mine[1] || hq[2] != mine[2] || hq[3] != mine[3])
            return fail(e, DK_ERR_INVALID_ARG, "accumulators differ between ranks %llu and %llu (units, unit capacity, record size or "
                        "window): create them with the same capacity and window on every rank", (unsigned long long)r, (unsigned long long)q);
        ovf_max = std::max(ovf_max, hq[5]);
        ovf_sum += hq[5];
    }
    dk_result *res = result_new(e);
    if (!res) return fail(e, DK_ERR_OOM, "host allocation failed");
    h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
    if (h != hipSuccess) { delete res; return fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h)); }
    memset(e->h_ctr, 0, sizeof(Counters));
    stage_begin(e);
    // 2. fill counters and stores: slice q of mine for slice r of rank q, in place
    const uint64_t upr = a->n_units / P;
    uint64_t sent = 0;
    a->exchanged = true;                          // from here on the store is no longer in unit order
    st = alltoall_in_place(e, (char *)a->fill, upr * 4, 4, &sent);
    if (st == DK_OK) st = alltoall_in_place(e, (char *)a->store, upr * (uint64_t)a->unit_cap * rb, 256, &sent);
    // 3. overflow lists (rare, small): padded all-gather, then packed side by side
    char *gath = nullptr, *extra = nullptr;
    if (st == DK_OK && ovf_max) {
        st = pool_alloc(e, P * ovf_max * ob, (void **)&gath);
        if (st == DK_OK) st = pool_alloc(e, ovf_sum * ob, (void **)&extra);
        if (st == DK_OK && my_ovf) {
            h = hipMemcpyAsync(gath + r * ovf_max * ob, a->ovf, my_ovf * ob, hipMemcpyDeviceToDevice, e->stream);
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow-list copy failed: %s", hipGetErrorString(h));
        }
        if (st == DK_OK) {
            const int rr = api->AllGather(gath + r * ovf_max * ob, gath, ovf_max * ob, RCCL_UINT8, c->comm, e->stream);
            if (rr != RCCL_SUCCESS) st = comm_fail(e, false, "ncclAllGather(overflow lists)", rr, __FILE__, __LINE__);
            else sent += (P - 1) * ovf_max * ob;
        }
        uint64_t done = 0;
        for (uint64_t q = 0; q < P && st == DK_OK; q++) {
            const uint64_t nq = hdr[q * HW + 5];
            if (!nq) continue;
            h = hipMemcpyAsync(extra + done * ob, gath + q * ovf_max * ob, nq * ob, hipMemcpyDeviceToDevice, e->stream);
            if (h != hipSuccess) st = fail(e, DK_ERR_HIP, "overflow-list copy failed: %s", hipGetErrorString(h));
            done += nq;
        }
    }
    if (st == DK_OK) stage_mark(e, "acc_exchange");
    // 4. count this rank's share of the units from the P pieces
    uint64_t n_records = 0;
    if (st == DK_OK)
        st = a->wide ? accum_finish_t<true>(e, a, min_count, res, a->store, a->fill, (uint32_t)P, r * upr, upr, extra, ovf_sum, &n_records)
                     : accum_finish_t<false>(e, a, min_count, res, a->store, a->fill, (uint32_t)P, r * upr, upr, extra, ovf_sum, &n_records);
    if (st == DK_OK) st = stage_end(e);
    pool_free(e, gath);
    pool_free(e, extra);
    if (st != DK_OK) { dk_result_destroy(res); return st; }
    if (bytes_sent) *bytes_sent = sent;
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_reads = a->n_reads;
        stats->n_bases = a->n_bases;
        stats->n_windows = a->n_windows;
        stats->n_valid = a->n_valid;
        stats->n_absent = n_records;             // the occurrences this rank counted (its share of the hash space, all ranks' reads)
        stats->n_distinct = e->h_ctr->n_distinct;
        stats->n_emitted = res->n;
    }
    *out = res;
    return DK_OK;
}

// host-only arithmetic of the exchanges, exported so that it can be checked without a GPU (tests/test_abi.py)
dk_status dk_comm_layout(uint64_t staging_bytes, uint64_t slice_bytes, uint32_t rank, uint32_t world_size, uint64_t granule,
                         uint64_t *piece_bytes, uint64_t *staging_slot)
{
    if (world_size < 1 || rank >= world_size || granule == 0 || !piece_bytes || !staging_slot) return DK_ERR_INVALID_ARG;
    *piece_bytes = exchange_piece_bytes(staging_bytes, slice_bytes, world_size, granule);
    for (uint32_t q = 0; q < world_size; q++)
        staging_slot[q] = q == rank && world_size > 1 ? ~0ULL : staging_index(q, rank, world_size);
    return DK_OK;
}

dk_status dk_result_size(const dk_result *res, uint64_t *n)
{
    if (!res || !n) return DK_ERR_INVALID_ARG;
    *n = res->n;
    return DK_OK;
}

dk_status dk_result_copy(const dk_result *res, uint64_t *kmers_lo, uint64_t *kmers_hi, uint32_t *counts)
{
    if (!res) return DK_ERR_INVALID_ARG;
    dk_engine *e = res->e;
    if (res->n == 0) return DK_OK;
    CHECK_ARG(e, kmers_lo != nullptr && counts != nullptr);
    CHECK_ARG(e, kmers_hi != nullptr || !res->wide);
    DK_HIP(e, hipSetDevice(e->device));
    uint64_t done = 0;
    for (uint32_t r = 0; r < res->n_regions; r++) {          // stitch the regions into dense host arrays
        const uint64_t cnt = res->region_n[r], src = (uint64_t)r * res->region_cap;
        if (!cnt) continue;
        // hipMemcpyDefault: the destinations may be host or device memory
        DK_HIP(e, hipMemcpyAsync(kmers_lo + done, res->d_lo + src, cnt * 8, hipMemcpyDefault, e->stream));
        if (kmers_hi && res->wide)
            DK_HIP(e, hipMemcpyAsync(kmers_hi + done, res->d_hi + src, cnt * 8, hipMemcpyDefault, e->stream));
        DK_HIP(e, hipMemcpyAsync(counts + done, res->d_cnt + src, cnt * 4, hipMemcpyDefault, e->stream));
        done += cnt;
    }
    if (kmers_hi && !res->wide) {
        if (is_device_ptr(kmers_hi)) DK_HIP(e, hipMemsetAsync(kmers_hi, 0, res->n * 8, e->stream));
        else memset(kmers_hi, 0, res->n * 8);
    }
    DK_HIP(e, hipStreamSynchronize(e->stream));
    return DK_OK;
}

dk_status dk_result_device_view(const dk_result *cres, const void **d_kmers_lo, const void **d_kmers_hi,
                                const void **d_counts, uint64_t *n)
{
    if (!cres) return DK_ERR_INVALID_ARG;
    dk_result *res = const_cast<dk_result *>(cres);
    dk_engine *e = res->e;
    if (res->n_regions > 1 && res->n) {
        // first device view of a regioned result: compact it into dense arrays (device-to-device)
        DK_HIP(e, hipSetDevice(e->device));
        uint64_t *lo = nullptr, *hi = nullptr;
        uint32_t *cnt = nullptr;
        dk_status st = pool_alloc(e, res->n * 8, (void **)&lo);
        if (st == DK_OK && res->wide) st = pool_alloc(e, res->n * 8, (void **)&hi);
        if (st == DK_OK) st = pool_alloc(e, res->n * 4, (void **)&cnt);
        if (st != DK_OK) { pool_free(e, lo); pool_free(e, hi); pool_free(e, cnt); return st; }
        uint64_t done = 0;
        hipError_t h = hipSuccess;
        for (uint32_t r = 0; r < res->n_regions && h == hipSuccess; r++) {
            const uint64_t c = res->region_n[r], src = (uint64_t)r * res->region_cap;
            if (!c) continue;
            h = hipMemcpyAsync(lo + done, res->d_lo + src, c * 8, hipMemcpyDeviceToDevice, e->stream);
            if (h == hipSuccess && res->wide) h = hipMemcpyAsync(hi + done, res->d_hi + src, c * 8, hipMemcpyDeviceToDevice, e->stream);
            if (h == hipSuccess) h = hipMemcpyAsync(cnt + done, res->d_cnt + src, c * 4, hipMemcpyDeviceToDevice, e->stream);
            done += c;
        }
        if (h == hipSuccess) h = hipStreamSynchronize(e->stream);
        if (h != hipSuccess) {
            pool_free(e, lo);
            pool_free(e, hi);
            pool_free(e, cnt);
            return fail(e, DK_ERR_HIP, "compacting result regions failed: %s", hipGetErrorString(h));
        }
        pool_free(e, res->d_lo);
        pool_free(e, res->d_hi);
        pool_free(e, res->d_cnt);
        res->d_lo = lo;
        res->d_hi = hi;
        res->d_cnt = cnt;
        res->n_regions = 1;
        res->region_cap = res->n;
        res->region_n[0] = res->n;
    }
    if (d_kmers_lo) *d_kmers_lo = res->d_lo;
    if (d_kmers_hi) *d_kmers_hi = res->d_hi;
    if (d_counts) *d_counts = res->d_cnt;
    if (n) *n = res->n;
    return DK_OK;
}

dk_status dk_result_merge(dk_engine *e, const dk_result *const *results, uint32_t n_results,
                          uint32_t min_count, dk_result **out, dk_stats *stats)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr && (results != nullptr || n_results == 0));
    CHECK_ARG(e, min_count >= 1);
    *out = nullptr;
    const bool wide = e->cfg.k > 32;
    for (uint32_t i = 0; i < n_results; i++) CHECK_ARG(e, results[i] != nullptr && results[i]->e == e && results[i]->wide == wide);
    DK_HIP(e, hipSetDevice(e->device));
    dk_result *res = new (std::nothrow) dk_result();
    if (!res) return fail(e, DK_ERR_OOM, "host allocation failed");
    res->e = e;
    res->d_lo = res->d_hi = nullptr;
    res->d_cnt = nullptr;
    res->n = 0;
    res->n_regions = 1;
    res->region_cap = 0;
    memset(res->region_n, 0, sizeof res->region_n);
    res->wide = wide;
    res->owns = true;
    hipError_t h = hipMemsetAsync(e->d_ctr, 0, sizeof(Counters), e->stream);
    if (h != hipSuccess) { delete res; return fail(e, DK_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(h)); }
    memset(e->h_ctr, 0, sizeof(Counters));
    stage_begin(e);
    dk_status st = wide ? merge_results<true>(e, results, n_results, min_count, res)
                        : merge_results<false>(e, results, n_results, min_count, res);
    if (st == DK_OK) st = stage_end(e);
    if (st != DK_OK) { dk_result_destroy(res); return st; }
    if (stats) {
        memset(stats, 0, sizeof *stats);
        stats->n_distinct = e->h_ctr->n_distinct;
        stats->n_emitted = res->n;
    }
    *out = res;
    return DK_OK;
}

dk_status dk_result_attach(dk_engine *e, const void *d_kmers_lo, const void *d_kmers_hi, const void *d_counts, uint64_t n,
                           dk_result **out)
{
    if (!e) return DK_ERR_INVALID_ARG;
    CHECK_ARG(e, out != nullptr);
    *out = nullptr;
    const bool wide = e->cfg.k > 32;
    CHECK_ARG(e, n == 0 || (d_kmers_lo && d_counts && (d_kmers_hi || !wide)));
    dk_result *res = new (std::nothrow) dk_result();
    if (!res) return fail(e, DK_ERR_OOM, "host allocation failed");
    res->e = e;
    res->d_lo = (uint64_t *)d_kmers_lo;
    res->d_hi = wide ? (uint64_t *)d_kmers_hi : nullptr;
    res->d_cnt = (uint32_t *)d_counts;
    res->n = n;
    res->wide = wide;
    res->owns = false;
    res->n_regions = 1;
    res->region_cap = n;
    memset(res->region_n, 0, sizeof res->region_n);
    res->region_n[0] = n;
    *out = res;
    return DK_OK;
}

void dk_result_destroy(dk_result *res)
{
    if (!res) return;
    if (res->owns) {
        pool_free(res->e, res->d_lo);
        pool_free(res->e, res->d_hi);
        pool_free(res->e, res->d_cnt);
    }
    delete res;
}

}  // extern "C"
