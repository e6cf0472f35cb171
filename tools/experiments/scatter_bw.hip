// scatter_bw.hip -- what HBM bandwidth does the multisplit WRITE PATTERN allow on its own?
// Each block copies one tile of 8192 8-byte records; record i of tile t goes to region (i / run) % nbins
// at slot t * run + i % run: runs of `run` records (run * 8 bytes contiguous) spread over nbins regions,
// exactly the shape repart / scan_part produce, with no LDS work, no atomics.  run = 8192 is a plain copy.
//   hipcc --offload-arch=gfx950 -O3 -o scatter_bw scatter_bw.hip && ./scatter_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

// WRITE_ONLY: the records are made up in registers -- the write side of the pattern alone (scan_part reads 1/32 of what
// it writes)
template <bool WRITE_ONLY>
__global__ void __launch_bounds__(1024) scatter_kernel(const uint64_t *__restrict__ in, uint64_t *__restrict__ out,
                                                        uint32_t run, uint32_t nbins, uint64_t region_stride)
{
    const uint64_t t = blockIdx.x;
    const uint64_t *src = in + t * 8192;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t i = j * 1024 + threadIdx.x;
        const uint64_t v = WRITE_ONLY ? t * 8192 + i : src[i];
        const uint32_t r = i / run, bin = r % nbins, rep = r / nbins;      // rep-th run of this tile in that bin
        const uint64_t runs_per_tile_bin = (8192 / run + nbins - 1) / nbins;
        out[bin * region_stride + (t * runs_per_tile_bin + rep) * run + i % run] = v;
    }
}

int main()
{
    const uint64_t n = 1536ULL << 20;            // 1.61 G records = 12.9 GB
    const uint32_t n_tiles = (uint32_t)(n / 8192);
    uint64_t *in, *out;
    hipMalloc(&in, n * 8);
    hipMalloc(&out, (n + (64ULL << 20)) * 8 * 2);
    hipMemset(in, 1, n * 8);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const uint32_t runs[] = {8192, 256, 128, 64, 32, 16, 8};
    for (uint32_t nbins : {64u, 256u, 1024u}) {
        for (uint32_t run : runs) {
            if (8192 / run < 1) continue;
            const uint64_t runs_per_tile_bin = (8192 / run + nbins - 1) / nbins;
            // skew: the regions' strides are otherwise multiples of 8 MiB, and all write frontiers advance in step
            for (uint32_t skew : {0u, 16u, 48u}) {
            const uint64_t region_stride = (uint64_t)n_tiles * runs_per_tile_bin * run + skew;
            if (region_stride * nbins > (n + (64ULL << 20)) * 2) continue;
            float best = 1e9f, best_w = 1e9f;
            for (int it = 0; it < 3; it++) {
                hipEventRecord(a);
                scatter_kernel<false><<<n_tiles, 1024>>>(in, out, run, nbins, region_stride);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
                hipEventRecord(a);
                scatter_kernel<true><<<n_tiles, 1024>>>(in, out, run, nbins, region_stride);
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms, a, b);
                if (ms < best_w) best_w = ms;
            }
            printf("nbins %6u run %5u records (%6u B) skew %3u B: %.2f ms  %.2f TB/s (read+write)   write only: %.2f ms  %.2f TB/s\n", nbins, run,
                   run * 8, skew * 8, best, 2.0 * n * 8 / best / 1e9, best_w, 1.0 * n * 8 / best_w / 1e9);
            }
        }
    }
    return 0;
}
