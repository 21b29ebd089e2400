// scopa_mt.h -- CPython's random.seed(int) + random.shuffle as the reference's decks use them (MiniDeck / FullDeck
// __init__: random.seed(seed); random.shuffle(cards)): MT19937 init_by_array on |seed|'s 32-bit words, then
// Random._randbelow_with_getrandbits for i = n-1 .. 1.  Host and device; on device the 624-word state sits in per-lane scratch
// and words are twisted on demand, in order.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace scopa {
struct Mt {
    uint32_t mt[624];
    int at;
    __host__ __device__ void seed(uint64_t a) {
        const uint32_t key[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
        const int klen = key[1] ? 2 : 1;
        mt[0] = 19650218u;
        for (int k = 1; k < 624; k++) mt[k] = 1812433253u * (mt[k - 1] ^ (mt[k - 1] >> 30)) + (uint32_t)k;
        int p = 1, j = 0;
        for (int k = 624; k > 0; k--) {
            mt[p] = (mt[p] ^ ((mt[p - 1] ^ (mt[p - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
            if (++p >= 624) { mt[0] = mt[623]; p = 1; }
            if (++j >= klen) j = 0;
        }
        for (int k = 623; k > 0; k--) {
            mt[p] = (mt[p] ^ ((mt[p - 1] ^ (mt[p - 1] >> 30)) * 1566083941u)) - (uint32_t)p;
            if (++p >= 624) { mt[0] = mt[623]; p = 1; }
        }
        mt[0] = 0x80000000u;
        at = 0;
    }
    __host__ __device__ uint32_t next() {
        const int k = at % 624;
        const uint32_t y0 = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y0 >> 1) ^ ((y0 & 1u) ? 0x9908b0dfu : 0u);
        uint32_t y = mt[k];
        at++;
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    __host__ __device__ void shuffle(uint8_t *perm, int n) {
        for (int i = 0; i < n; i++) perm[i] = (uint8_t)i;
        for (int hi = n - 1; hi >= 1; hi--) {
            const uint32_t nn = (uint32_t)hi + 1u;
            int bits = 0;
            for (uint32_t t = nn; t; t >>= 1) bits++;
            uint32_t r;
            do r = next() >> (32 - bits); while (r >= nn);
            const uint8_t t = perm[hi]; perm[hi] = perm[r]; perm[r] = t;
        }
    }
};
}  // namespace scopa
