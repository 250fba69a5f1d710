// odw_rng.h -- device twin of od_wscl_amd/utils/rng.py (counter-based, bit-identical
// uniform stream; the numpy side is the checker's copy).
#pragma once
#include <stdint.h>

__device__ __forceinline__ uint32_t odw_mix(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t odw_bits(uint32_t idx, uint32_t k0, uint32_t k1) {
    return odw_mix(odw_mix(idx ^ k0) + k1);
}
// [0,1) with 24 random bits
__device__ __forceinline__ float odw_uniform(uint32_t idx, uint32_t k0, uint32_t k1) {
    return (float)(odw_bits(idx, k0, k1) >> 8) * (1.0f / 16777216.0f);
}
