// Does gfx950 execute scalar-memory atomics (s_atomic_add, result through lgkmcnt -- not behind the wave's vector stores)?
//   hipcc --offload-arch=gfx950 -O3 -o scalar_atomic scalar_atomic.hip && ./scalar_atomic
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *ctr, unsigned *out)
{
    unsigned r = 3u;                                   // every wave adds 3 and gets the old value
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n s_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(ctr) : "memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = r;
}
int main()
{
    unsigned *ctr, *out, h[4096], hc;
    hipMalloc(&ctr, 4);
    hipMalloc(&out, sizeof h);
    hipMemset(ctr, 0, 4);
    k<<<256, 1024>>>(ctr, out);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    hipMemcpy(&hc, ctr, 4, hipMemcpyDeviceToHost);
    unsigned long long sum = 0;
    unsigned mx = 0;
    for (int i = 0; i < 4096; i++) { sum += h[i]; if (h[i] > mx) mx = h[i]; }
    // 4096 waves x 3: counter 12288; old values = 0, 3, ..., 12285 in some order: sum 3 * 4095 * 4096 / 2
    printf("%s counter %u (want 12288) sum_old %llu (want %llu) max_old %u\n", hipGetErrorString(e), hc, sum, 3ULL * 4095 * 4096 / 2, mx);
    return 0;
}
