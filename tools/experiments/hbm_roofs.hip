// tools/experiments/hbm_roofs.hip -- what plain streaming kernels reach on this GPU: read-only, write-only, copy and a 50/50
// mix over buffers far larger than the caches.  The pipeline's kernels are priced against the 8 TB/s peak (bench.py's
// roofline.peak); these figures say how much of that peak ANY kernel can have.
//   hipcc --offload-arch=gfx950 -O3 -o hbm_roofs hbm_roofs.hip && ./hbm_roofs [GiB per buffer, default 8]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool NT>
__global__ void __launch_bounds__(1024) read_kernel(const uint4 *__restrict__ src, size_t n, uint64_t *sink)
{
    uint64_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v;
        if (NT) {
            v.x = __builtin_nontemporal_load(&src[i].x); v.y = __builtin_nontemporal_load(&src[i].y);
            v.z = __builtin_nontemporal_load(&src[i].z); v.w = __builtin_nontemporal_load(&src[i].w);
        } else v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x1234567) *sink = acc;
}

template <bool NT>
__global__ void __launch_bounds__(1024) write_kernel(uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t x = (uint32_t)i;
        if (NT) {
            __builtin_nontemporal_store(x, &dst[i].x); __builtin_nontemporal_store(x + 1, &dst[i].y);
            __builtin_nontemporal_store(x + 2, &dst[i].z); __builtin_nontemporal_store(x + 3, &dst[i].w);
        } else dst[i] = make_uint4(x, x + 1, x + 2, x + 3);
    }
}

__global__ void __launch_bounds__(1024) copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// two reads per write (a set kernel's mix: records + segment in, a smaller stream out)
__global__ void __launch_bounds__(1024) read2_write1_kernel(const uint4 *__restrict__ a, const uint4 *__restrict__ b, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 x = a[i], y = b[i];
        dst[i] = make_uint4(x.x ^ y.x, x.y ^ y.y, x.z ^ y.z, x.w ^ y.w);
    }
}

template <class F>
static double time_ms(F f, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char **argv)
{
    const size_t gib = argc > 1 ? (size_t)atoi(argv[1]) : 8;
    const size_t bytes = gib << 30, n = bytes / 16;
    uint4 *a, *b, *c;
    uint64_t *sink;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&c, bytes));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(a, 1, bytes));
    CK(hipMemset(b, 2, bytes));
    CK(hipMemset(c, 3, bytes));
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("{\"device\": \"%s\", \"cus\": %d, \"gib_per_buffer\": %zu", p.name, p.multiProcessorCount, gib);
    for (int per_cu : {2, 8}) {
        const unsigned grid = (unsigned)p.multiProcessorCount * per_cu;
        auto gbs = [&](double ms, double nbuf) { return nbuf * bytes / (ms * 1e-3) / 1e9; };
        double ms;
        ms = time_ms([&] { read_kernel<false><<<grid, 1024>>>(a, n, sink); }, 5);
        printf(", \"read_%dwg\": %.0f", per_cu, gbs(ms, 1));
        ms = time_ms([&] { read_kernel<true><<<grid, 1024>>>(a, n, sink); }, 5);
        printf(", \"read_nt_%dwg\": %.0f", per_cu, gbs(ms, 1));
        ms = time_ms([&] { write_kernel<false><<<grid, 1024>>>(c, n); }, 5);
        printf(", \"write_%dwg\": %.0f", per_cu, gbs(ms, 1));
        ms = time_ms([&] { write_kernel<true><<<grid, 1024>>>(c, n); }, 5);
        printf(", \"write_nt_%dwg\": %.0f", per_cu, gbs(ms, 1));
        ms = time_ms([&] { copy_kernel<<<grid, 1024>>>(a, c, n); }, 5);
        printf(", \"copy_%dwg\": %.0f", per_cu, gbs(ms, 2));
        ms = time_ms([&] { read2_write1_kernel<<<grid, 1024>>>(a, b, c, n); }, 5);
        printf(", \"read2_write1_%dwg\": %.0f", per_cu, gbs(ms, 3));
    }
    double ms = time_ms([&] { CK(hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 5);
    printf(", \"hipMemcpyDtoD\": %.0f", 2.0 * bytes / (ms * 1e-3) / 1e9);
    ms = time_ms([&] { CK(hipMemsetAsync(c, 0, bytes, 0)); }, 5);
    printf(", \"hipMemset\": %.0f", 1.0 * bytes / (ms * 1e-3) / 1e9);
    printf(", \"unit\": \"GB/s\"}\n");
    return 0;
}
