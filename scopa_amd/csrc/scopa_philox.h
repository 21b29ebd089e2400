// scopa_philox.h -- Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw: "Parallel random numbers:
// as easy as 1, 2, 3", SC'11), own implementation so that every device, rank and launch split draws the same
// numbers.  Known-answer vectors of the Random123 distribution are checked in tests/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scopa {

struct philox_out { uint32_t x0, x1, x2, x3; };

__host__ __device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

__host__ __device__ __forceinline__ philox_out philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                              uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        // one 32x32->64 multiply per product (v_mad_u64_u32 on gfx950) instead of separate mul_hi / mul_lo
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
        const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
        const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
        c0 = n0; c1 = l1; c2 = n2; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return philox_out{c0, c1, c2, c3};
}

// 53-bit uniform in [0,1) from two 32-bit words: the genrand_res53 construction behind numpy's random_sample
__host__ __device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) / 9007199254740992.0;
}

}  // namespace scopa
