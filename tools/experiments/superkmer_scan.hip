// superkmer_scan.hip -- how fast can the scan of the planned super-k-mer layout run?
// Each thread owns 32 consecutive k-mer start positions of a 2-bit packed stream (word aligned: three
// bases words and two mask words cover everything it needs), computes the canonical m-mer order keys of
// the 48 m-mers involved (fmix32 of the canonical m-mer, packed with the local offset), takes the sliding
// minimum over w = k-m+1 keys with a doubling tree (5 v_min_u32 per position), cuts super-k-mers where
// the minimizer position changes or a window is not a k-mer, and emits one 16-byte record per super-k-mer
// {route:18 | count:6 | bases:40}{bases:64}, compacted per wave.  No partitioning: this measures the
// extraction side only.  A CPU loop re-derives the number of records and k-mers for a slice.
//   hipcc --offload-arch=gfx950 -O3 -o superkmer_scan superkmer_scan.hip && ./superkmer_scan
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

constexpr int K = 31, M = 15, W = K - M + 1;          // 17 m-mers per k-mer
constexpr int PT = 32;                                // k-mer starts per thread
constexpr int NM = PT + W - 1;                        // 48 m-mers per thread
constexpr uint32_t MMASK = (1u << (2 * M)) - 1;

__host__ __device__ inline uint32_t fmix32(uint32_t x)
{
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}

// base i (0 = first base of the thread's chunk) of the 96 bases held MSB-first in w0, w1, w2
__host__ __device__ inline uint32_t base_at(uint64_t w0, uint64_t w1, uint64_t w2, int i)
{
    const uint64_t w = i < 32 ? w0 : i < 64 ? w1 : w2;
    return (uint32_t)(w >> (62 - 2 * (i & 31))) & 3u;
}

struct Rec { uint64_t a, b; };

__global__ void __launch_bounds__(256)
superkmer_kernel(const uint64_t *__restrict__ bases, const uint64_t *__restrict__ mask, uint64_t n_pos, uint64_t n_words,
                 Rec *__restrict__ out, unsigned long long *counters)
{
    unsigned long long tot_rec = 0, tot_kmers = 0;
    const uint64_t n_threads_total = (n_pos + PT - 1) / PT;
    for (uint64_t gt = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gt < ((n_threads_total + 63) & ~63ULL);
         gt += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t p0 = gt * PT;
    const bool active = p0 < n_pos;
    const uint64_t wi = p0 >> 5;                                               // PT = 32: word aligned
    const uint64_t last = n_words - 1;
    const uint64_t w0 = bases[wi < last ? wi : last], w1 = bases[wi + 1 < last ? wi + 1 : last],
                   w2 = bases[wi + 2 < last ? wi + 2 : last];
    // mask: 1 = not a base (N / separator); 96 flags from two 64-bit words
    const uint64_t mi = p0 >> 6, mlast = (n_words + 1) / 2 - 1;
    const uint64_t mw0 = mask[mi < mlast ? mi : mlast], mw1 = mask[mi + 1 < mlast ? mi + 1 : mlast],
                   mw2 = mask[mi + 2 < mlast ? mi + 2 : mlast];
    const int ms = (int)(p0 & 63);                                             // 0 or 32
    const uint64_t mh = ms ? (mw0 << 32) | (mw1 >> 32) : mw0;                  // flags of positions p0 .. p0+63
    const uint64_t ml = ms ? (mw1 << 32) | (mw2 >> 32) : mw1;                  // p0+64 .. p0+127

    // order keys of the m-mers starting at 0 .. NM-1
    uint32_t key[NM];
    uint32_t fwd = 0, rc = 0;
#pragma unroll
    for (int i = 0; i < M - 1; i++) {
        const uint32_t b = base_at(w0, w1, w2, i);
        fwd = (fwd << 2) | b;
        rc = (rc >> 2) | ((3u - b) << (2 * (M - 1)));
    }
#pragma unroll
    for (int i = 0; i < NM; i++) {
        const uint32_t b = base_at(w0, w1, w2, i + M - 1);
        fwd = ((fwd << 2) | b) & MMASK;
        rc = (rc >> 2) | ((3u - b) << (2 * (M - 1)));
        const uint32_t can = fwd < rc ? fwd : rc;
        key[i] = (fmix32(can) & ~63u) | (uint32_t)i;
    }
    // sliding minimum over W = 17 keys: doubling tree
#pragma unroll
    for (int i = 0; i + 1 < NM; i++) key[i] = key[i] < key[i + 1] ? key[i] : key[i + 1];               // spans 2
    // careful: in-place doubling needs the previous level's values at i + span; go level by level on copies
    uint32_t m2[NM], m4[NM], m8[NM], m16[NM];
#pragma unroll
    for (int i = 0; i < NM; i++) m2[i] = key[i];
    // (key[] above already holds min over [i, i+1]; recompute the original key of the 17th element below)
#pragma unroll
    for (int i = 0; i + 2 < NM; i++) m4[i] = m2[i] < m2[i + 2] ? m2[i] : m2[i + 2];
#pragma unroll
    for (int i = 0; i + 4 < NM - 2; i++) m8[i] = m4[i] < m4[i + 4] ? m4[i] : m4[i + 4];
#pragma unroll
    for (int i = 0; i + 8 < NM - 6; i++) m16[i] = m8[i] < m8[i + 8] ? m8[i] : m8[i + 8];
    // min over 17 = min(min16 over [j, j+15], min2 over [j+15, j+16])
    uint32_t minpos[PT];
    uint64_t notk = 0;                                                          // bit j: window j is not a k-mer
#pragma unroll
    for (int j = 0; j < PT; j++) {
        const uint32_t v = m16[j] < m2[j + 15] ? m16[j] : m2[j + 15];
        minpos[j] = v & 63u;
        // k flags from position j: all must be zero
        const uint64_t f = j ? (mh << j) | (ml >> (64 - j)) : mh;
        const bool bad = (f >> (64 - K)) != 0 || p0 + j >= n_pos;
        notk |= (uint64_t)bad << j;
    }
    // cut points and records
    uint32_t n_rec = 0, n_kmers = 0;
    uint32_t starts = 0;                                                        // bit j: a super-k-mer starts at j
#pragma unroll
    for (int j = 0; j < PT; j++) {
        const bool valid = !((notk >> j) & 1);
        const bool prev_valid = j > 0 && !((notk >> (j - 1)) & 1);
        const bool cut = valid && (!prev_valid || minpos[j] != minpos[j - (j > 0)]);
        starts |= (uint32_t)cut << j;
        n_kmers += valid;
    }
    if (!active) starts = 0;
    n_rec = __popc(starts);
    // wave-level compaction of the records
    uint32_t incl = n_rec;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(incl, d);
        if ((int)(threadIdx.x & 63) >= d) incl += t;
    }
    const uint32_t wave_total = __shfl(incl, 63);
    // every wave tile owns a fixed output slot of 64 * PT / 3 records (a shared append counter would cap the
    // kernel near 10^8 waves per second); the real kernel hands its records to the LDS multisplit instead
    constexpr uint32_t SLOT = 64 * PT / 3;
    const uint64_t base = (gt >> 6) * SLOT;
    uint64_t o = base + incl - n_rec;
    if ((threadIdx.x & 63) == 63) tot_rec += wave_total;
    uint32_t rest = starts;
    while (rest) {
        const int j = __ffs(rest) - 1;
        rest &= rest - 1;
        // length: up to the next start or the first invalid window
        const uint32_t after = (starts >> 1 >> j) << 1 << j;                     // starts above j
        const uint64_t inval_after = (notk >> j) << j;
        int end = PT;
        if (after) end = __ffs(after) - 1;
        if (inval_after) { const int e2 = __ffsll((long long)inval_after) - 1; end = e2 < end ? e2 : end; }
        const uint32_t count = (uint32_t)(end - j);
        // bases j .. j + count - 1 + K - 1 (<= 47 + ... bases): take 52 bases from position j
        const int sh = 2 * j;                                                    // 0..62
        const uint64_t x0 = sh ? (w0 << sh) | (w1 >> (64 - sh)) : w0;
        const uint64_t x1 = sh ? (w1 << sh) | (w2 >> (64 - sh)) : w1;
        // route: independent mix of the canonical minimizer
        uint32_t mf = 0;
        const int mp = (int)minpos[j];
#pragma unroll 1
        for (int i = 0; i < M; i++) mf = (mf << 2) | base_at(w0, w1, w2, mp + i);
        uint32_t mr = 0;
#pragma unroll 1
        for (int i = 0; i < M; i++) mr |= (3u - ((mf >> (2 * i)) & 3u)) << (2 * (M - 1 - i));
        const uint32_t can = mf < mr ? mf : mr;
        const uint32_t route = fmix32(can ^ 0x9E3779B9u) >> 14;                  // 18 bits
        Rec r;
        r.a = ((uint64_t)route << 46) | ((uint64_t)count << 40) | (x0 >> 24);
        r.b = (x0 << 40) | (x1 >> 24);
        if (o - base < SLOT) out[o] = r;
        o++;
    }
    tot_kmers += n_kmers;
    }
    for (int d = 32; d > 0; d >>= 1) tot_kmers += __shfl_down(tot_kmers, d);
    if ((threadIdx.x & 63) == 0 && tot_kmers) atomicAdd(&counters[1], tot_kmers);
    if ((threadIdx.x & 63) == 63 && tot_rec) atomicAdd(&counters[0], tot_rec);
}

// ---- segment-kernel side: expand the k-mers of each record, hash them, test four bits of an LDS-resident filter ----
__device__ __forceinline__ uint64_t fmix64(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
__device__ __forceinline__ uint64_t rev_pairs64(uint64_t x)
{
    uint64_t r = __brevll(x);
    return ((r >> 1) & 0x5555555555555555ULL) | ((r & 0x5555555555555555ULL) << 1);
}
// k-mer number idx of a record: bases idx .. idx + K - 1 of the 104-bit base string {a[39:0], b}
__device__ __forceinline__ uint64_t kmer_of(const Rec &r, uint32_t idx)
{
    const uint64_t hi = r.a << 24 | r.b >> 40, lo = r.b << 24;                   // bases left-aligned in (hi, lo)
    const int sh = 2 * (int)idx;
    const uint64_t v = sh ? (hi << sh) | (lo >> (64 - sh)) : hi;
    const uint64_t fwd = v >> (64 - 2 * K);
    const uint64_t rc = (~rev_pairs64(fwd)) >> (64 - 2 * K);
    return rc < fwd ? rc : fwd;
}
__device__ __forceinline__ bool test4(const uint32_t *seg, uint64_t h)
{
    const uint32_t blk = (uint32_t)(h >> 39) & 1023u, a = (uint32_t)(h & 511), d = (uint32_t)((h >> 9) & 511) | 1u;
    uint32_t acc = 1u;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint32_t bit = (a + (uint32_t)j * d) & 511;
        acc &= seg[blk * 16 + (bit >> 5)] >> (bit & 31);
    }
    return acc & 1u;
}

// A: one record per lane, each lane loops over its own k-mers (the wave runs to its longest record)
__global__ void __launch_bounds__(1024)
expand_lane_kernel(const Rec *__restrict__ recs, uint64_t n_slots, unsigned long long *counters)
{
    __shared__ uint32_t seg[16384];
    for (int i = threadIdx.x; i < 16384; i += 1024) seg[i] = fmix32(i * 2654435761u + blockIdx.x);
    __syncthreads();
    unsigned long long hits = 0, kmers = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n_slots; i += (uint64_t)gridDim.x * 1024) {
        const Rec r = recs[i];
        const uint32_t count = (uint32_t)(r.a >> 40) & 63u;
        for (uint32_t c = 0; c < count; c++) hits += test4(seg, fmix64(kmer_of(r, c) ^ 20260313ULL));
        kmers += count;
    }
    for (int d = 32; d > 0; d >>= 1) { hits += __shfl_down(hits, d); kmers += __shfl_down(kmers, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counters[2], hits); atomicAdd(&counters[3], kmers); }
}

// B: the k-mers of a wave's 64 records are flattened: lane l takes k-mers l, l + 64, ... of the wave's list
__global__ void __launch_bounds__(1024)
expand_flat_kernel(const Rec *__restrict__ recs, uint64_t n_slots, unsigned long long *counters)
{
    __shared__ uint32_t seg[16384];
    __shared__ Rec wrec[16][64];
    __shared__ uint32_t wpre[16][65];
    for (int i = threadIdx.x; i < 16384; i += 1024) seg[i] = fmix32(i * 2654435761u + blockIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    unsigned long long hits = 0, kmers = 0;
    const uint64_t n_round = (n_slots + 63) & ~63ULL;
    for (uint64_t i = (uint64_t)blockIdx.x * 1024 + threadIdx.x; i < n_round; i += (uint64_t)gridDim.x * 1024) {
        Rec r{0, 0};
        if (i < n_slots) r = recs[i];
        const uint32_t count = (uint32_t)(r.a >> 40) & 63u;
        uint32_t incl = count;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        wrec[wv][lane] = r;
        wpre[wv][lane + 1] = incl;
        if (lane == 0) wpre[wv][0] = 0;
        const uint32_t total = __shfl(incl, 63);
        for (uint32_t t = lane; t < total; t += 64) {
            // record of k-mer t: largest q with wpre[q] <= t
            uint32_t q = 0;
#pragma unroll
            for (int step = 32; step > 0; step >>= 1)
                if (wpre[wv][q + step] <= t) q += step;
            const Rec rr = wrec[wv][q];
            hits += test4(seg, fmix64(kmer_of(rr, t - wpre[wv][q]) ^ 20260313ULL));
        }
        kmers += count;
    }
    for (int d = 32; d > 0; d >>= 1) { hits += __shfl_down(hits, d); kmers += __shfl_down(kmers, d); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counters[2], hits); atomicAdd(&counters[3], kmers); }
}

int main()
{
    const uint64_t n_reads = 12800000, L = 150;
    const uint64_t n_pos = n_reads * (L + 1);
    const uint64_t n_words = (n_pos + 31) / 32 + 4;
    std::vector<uint64_t> hb(n_words), hm((n_words + 1) / 2 + 2, 0);
    uint64_t s = 88172645463325252ULL;
    for (auto &w : hb) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = s; }
    for (uint64_t r = 0; r < n_reads; r++) {
        const uint64_t p = r * (L + 1) + L;                                      // separator
        hm[p >> 6] |= 1ULL << (63 - (p & 63));
    }
    uint64_t *db, *dm;
    Rec *dout;
    unsigned long long *dc;
    hipMalloc(&db, n_words * 8);
    hipMalloc(&dm, hm.size() * 8);
    hipMalloc(&dout, (n_pos / 3 + (1 << 20)) * sizeof(Rec));
    hipMalloc(&dc, 64);
    hipMemcpy(db, hb.data(), n_words * 8, hipMemcpyHostToDevice);
    hipMemcpy(dm, hm.data(), hm.size() * 8, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const uint64_t n_threads = (n_pos + PT - 1) / PT;
    const unsigned grid = (unsigned)(((n_threads + 255) / 256) < 256 * 8 ? (n_threads + 255) / 256 : 256 * 8);
    float best = 1e9f;
    unsigned long long hc[2] = {0, 0};
    for (int it = 0; it < 4; it++) {
        hipMemset(dc, 0, 64);
        hipEventRecord(a);
        superkmer_kernel<<<grid, 256>>>(db, dm, n_pos, n_words, dout, dc);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
        hipMemcpy(hc, dc, 16, hipMemcpyDeviceToHost);
    }
    printf("k=%d m=%d: %llu positions, %llu k-mers, %llu records (%.2f k-mers per record, %.2f B per k-mer)\n", K, M,
           (unsigned long long)n_pos, hc[1], hc[0], (double)hc[1] / hc[0], 16.0 * hc[0] / hc[1]);
    printf("scan: %.2f ms = %.1f Gk-mers/s; bytes in %.2f GB + out %.2f GB = %.0f GB/s\n", best, hc[1] / best / 1e6,
           n_pos * 3 / 8 / 1e9, hc[0] * 16 / 1e9, (n_pos * 3 / 8 + hc[0] * 16.0) / best / 1e6);
    // the output slots are 64 * PT / 3 records per wave tile; unused slots must read as count 0: redo the scan
    // into a zeroed buffer, then run the two expansion variants over all slots
    const uint64_t n_slots = ((n_threads + 63) / 64) * (64 * PT / 3);
    hipMemset(dout, 0, n_slots * sizeof(Rec));
    hipMemset(dc, 0, 64);
    superkmer_kernel<<<grid, 256>>>(db, dm, n_pos, n_words, dout, dc);
    hipDeviceSynchronize();
    for (int variant = 0; variant < 2; variant++) {
        float bestx = 1e9f;
        unsigned long long hx[4] = {0, 0, 0, 0};
        for (int it = 0; it < 3; it++) {
            hipMemset(dc, 0, 64);
            hipEventRecord(a);
            if (variant == 0) expand_lane_kernel<<<512, 1024>>>(dout, n_slots, dc);
            else expand_flat_kernel<<<512, 1024>>>(dout, n_slots, dc);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            if (ms < bestx) bestx = ms;
            hipMemcpy(hx, dc, 32, hipMemcpyDeviceToHost);
        }
        printf("expand (%s): %.2f ms = %.1f Gk-mers/s, %llu k-mers, %llu filter hits (slots read: %.2f GB)\n",
               variant == 0 ? "lane per record" : "flattened per wave", bestx, hx[3] / bestx / 1e6, hx[3], hx[2], n_slots * 16 / 1e9);
    }
    // CPU re-derivation of the counts on the first reads
    const uint64_t chk_reads = 20000;
    uint64_t c_rec = 0, c_km = 0;
    auto base_of = [&](uint64_t p) { return (uint32_t)(hb[p >> 5] >> (62 - 2 * (p & 31))) & 3u; };
    // the GPU cuts at multiples of PT positions as well
    for (uint64_t r = 0; r < chk_reads; r++) {
        const uint64_t p0 = r * (L + 1);
        long prev_min = -1;
        bool prev_valid = false;
        for (uint64_t j = 0; j + K <= L; j++) {
            uint32_t bestk = 0xFFFFFFFFu;
            long bestp = -1;
            for (int t = 0; t < W; t++) {
                uint32_t f = 0, rc = 0;
                for (int i = 0; i < M; i++) {
                    const uint32_t bb = base_of(p0 + j + t + i);
                    f = (f << 2) | bb;
                    rc |= (3u - bb) << (2 * i);
                }
                const uint32_t can = f < rc ? f : rc;
                const uint32_t hk = fmix32(can) & ~63u;
                if (hk < bestk) { bestk = hk; bestp = (long)(p0 + j + t); }
            }
            const bool chunk_start = ((p0 + j) % PT) == 0;
            if (!prev_valid || bestp != prev_min || chunk_start) c_rec++;
            prev_min = bestp;
            prev_valid = true;
            c_km++;
        }
    }
    printf("CPU check on %llu reads: %llu k-mers, %llu records (ties between equal hashes may differ slightly)\n",
           (unsigned long long)chk_reads, (unsigned long long)c_km, (unsigned long long)c_rec);
    return 0;
}
