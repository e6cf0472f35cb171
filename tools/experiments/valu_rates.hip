// valu_rates.hip -- issue rate of the vector instructions scan_part is made of, on this chip.
// Every wave runs a long unrolled chain of one instruction on 8 independent register groups (so latency is
// covered inside the wave), 8 waves per SIMD; the result is lane-operations per CU and clock, where a
// full-rate instruction reaches 64 (4 SIMDs x 16 lanes).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define REP8(x) x x x x x x x x
#define ITERS 512

#define KERNEL32(name, ASM)                                                                        \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t s0)                         \
    {                                                                                              \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < ITERS; i++) {                                                          \
            REP8(asm volatile(ASM : "+v"(a0) : "s"(s0)); asm volatile(ASM : "+v"(a1) : "s"(s0));    \
                 asm volatile(ASM : "+v"(a2) : "s"(s0)); asm volatile(ASM : "+v"(a3) : "s"(s0));    \
                 asm volatile(ASM : "+v"(a4) : "s"(s0)); asm volatile(ASM : "+v"(a5) : "s"(s0));    \
                 asm volatile(ASM : "+v"(a6) : "s"(s0)); asm volatile(ASM : "+v"(a7) : "s"(s0));)   \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;               \
    }

#define KERNEL64(name, ASM)                                                                        \
    __global__ void __launch_bounds__(256) name(uint32_t *out, uint32_t s0)                         \
    {                                                                                              \
        const uint32_t m = threadIdx.x * 2654435761u + 1u;                                       \
        uint64_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        for (int i = 0; i < ITERS; i++) {                                                          \
            REP8(asm volatile(ASM : "+v"(a0) : "s"(s0), "v"(m) : "vcc"); asm volatile(ASM : "+v"(a1) : "s"(s0), "v"(m) : "vcc");    \
                 asm volatile(ASM : "+v"(a2) : "s"(s0), "v"(m) : "vcc"); asm volatile(ASM : "+v"(a3) : "s"(s0), "v"(m) : "vcc");    \
                 asm volatile(ASM : "+v"(a4) : "s"(s0), "v"(m) : "vcc"); asm volatile(ASM : "+v"(a5) : "s"(s0), "v"(m) : "vcc");    \
                 asm volatile(ASM : "+v"(a6) : "s"(s0), "v"(m) : "vcc"); asm volatile(ASM : "+v"(a7) : "s"(s0), "v"(m) : "vcc");)   \
        }                                                                                          \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7);   \
    }

KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1")
KERNEL32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %0, %1")
KERNEL32(k_add3, "v_add3_u32 %0, %0, %1, %0")
KERNEL32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 3, %0")
KERNEL32(k_bfe_u32, "v_bfe_u32 %0, %0, %1, 9")
KERNEL64(k_lshrrev_b64, "v_lshrrev_b64 %0, %1, %0")
KERNEL64(k_lshlrev_b64, "v_lshlrev_b64 %0, %1, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 3, %0")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %2, %1, %0")
KERNEL64(k_cmp_lt_u64, "v_cmp_lt_u64 vcc, %0, %0")

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const double clock_hz = prop.clockRate * 1e3;
    const int blocks = n_cu * 8;                  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    uint32_t *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    printf("%d CUs at %.0f MHz (reported); lane-operations per CU and clock, full rate = 64\n", n_cu, clock_hz / 1e6);
#define RUN(k, per)                                                                                  \
    do {                                                                                             \
        k<<<blocks, 256>>>(out, 7);                                                                  \
        hipDeviceSynchronize();                                                                      \
        hipEventRecord(a);                                                                           \
        for (int r = 0; r < 5; r++) k<<<blocks, 256>>>(out, 7);                                      \
        hipEventRecord(b);                                                                           \
        hipEventSynchronize(b);                                                                      \
        float ms;                                                                                    \
        hipEventElapsedTime(&ms, a, b);                                                              \
        const double ops = 5.0 * blocks * 256.0 * ITERS * 64.0 * per;                                \
        printf("%-16s %6.1f lane-ops/CU/clk  (%.2f ms)\n", #k, ops / (ms * 1e-3) / n_cu / clock_hz, ms / 5); \
    } while (0)
    RUN(k_add_u32, 1);
    RUN(k_xor_b32, 1);
    RUN(k_add3, 1);
    RUN(k_lshl_add_u32, 1);
    RUN(k_bfe_u32, 1);
    RUN(k_alignbit, 1);
    RUN(k_mul_u32_u24, 1);
    RUN(k_mad_u32_u24, 1);
    RUN(k_mul_lo_u32, 1);
    RUN(k_mul_hi_u32, 1);
    RUN(k_mad_u64_u32, 1);
    RUN(k_lshrrev_b64, 1);
    RUN(k_lshlrev_b64, 1);
    RUN(k_lshl_add_u64, 1);
    RUN(k_cmp_lt_u64, 1);
    return 0;
}
