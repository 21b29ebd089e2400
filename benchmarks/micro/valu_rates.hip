// Issue cost of the vector instructions the traversal kernel leans on (gfx950): cycles per wave64 instruction when ONE wavefront per
// SIMD issues a long independent stream of them.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void __launch_bounds__(64) k(unsigned long long *out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u;
    uint64_t q0 = a0, q1 = a1, q2 = a2, q3 = a3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < 64; it++) {
        if (OP == 0) { REP64(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 1) { REP64(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %5, %6, %1\n v_mad_u64_u32 %2, vcc, %6, %7, %2\n v_mad_u64_u32 %3, vcc, %7, %4, %3" : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        if (OP == 2) { REP64(asm volatile("v_mul_lo_u32 %0, %0, %1\n v_mul_lo_u32 %1, %1, %2\n v_mul_lo_u32 %2, %2, %3\n v_mul_lo_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 3) { REP64(asm volatile("v_mul_u32_u24 %0, %0, %1\n v_mul_u32_u24 %1, %1, %2\n v_mul_u32_u24 %2, %2, %3\n v_mul_u32_u24 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (OP == 4) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %2, %2, %3, %0\n v_fma_f64 %3, %3, %0, %1" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 5) { REP64(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 6) { REP64(asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %1, %1, %2\n v_mul_f64 %2, %2, %3\n v_mul_f64 %3, %3, %0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));) }
        if (OP == 7) { REP64(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");) }
        if (OP == 8) { REP64(asm volatile("v_mul_hi_u32 %0, %0, %1\n v_mul_hi_u32 %1, %1, %2\n v_mul_hi_u32 %2, %2, %3\n v_mul_hi_u32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    const unsigned long long t1 = clock64();
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = a0 + a1 + a2 + a3 + (uint32_t)(q0 + q1 + q2 + q3) + (uint32_t)(d0 + d1 + d2 + d3);
}

template <int OP> static void run(const char *name, int waves_per_simd) {
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 4096);
    // one workgroup of 64 x (4 * waves_per_simd) threads would share a CU; separate 64-thread workgroups land anywhere: use 1 block = 1 wave, 1 block total
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, d, 1u);
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, d, 1u);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("%-16s %6.2f clocks per instruction (one wavefront, 4 independent chains)\n", name, (double)h[0] / (64.0 * 64 * 4));
    (void)hipFree(d);
}

int main() {
    run<0>("v_xor_b32", 1); run<3>("v_mul_u32_u24", 1); run<7>("v_cndmask_b32", 1); run<1>("v_mad_u64_u32", 1); run<2>("v_mul_lo_u32", 1); run<8>("v_mul_hi_u32", 1);
    run<4>("v_fma_f64", 1); run<6>("v_mul_f64", 1); run<5>("v_rcp_f64", 1);
    return 0;
}
