// Counter-based dropout mask shared by every fused kernel (restated in oracle/rng.py for the tests).
//   keep(seed, stream, idx) = fmix32(idx + stream*0x85EBCA77 + seed) >= floor(p * 2^32)
// (murmur3's finaliser over a counter: two 32-bit multiplies per element -- quarter-rate instructions on CDNA, and a mask bit is
// wanted for every element of five tensors per block; the stream / seed term is wave-uniform and costs nothing per element)
// Stream ids: 1 = embedding dropout; block l: 16l+2 attention probs, 16l+3 FFN dropout1, 16l+4 FFN dropout2.
#pragma once
#include <stdint.h>

#define RE_STREAM_EMBED 1u
#define RE_STREAM_ATTN(l) (16u * (l) + 2u)
#define RE_STREAM_FFN1(l) (16u * (l) + 3u)
#define RE_STREAM_FFN2(l) (16u * (l) + 4u)

__host__ __device__ __forceinline__ uint32_t re_rng_u32(uint32_t seed, uint32_t stream, uint32_t idx) {
    uint32_t h = idx + (stream * 0x85EBCA77u + seed);
    h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
    return h;
}
static inline uint32_t re_drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    return t >= 4294967295.0 ? 4294967295u : (uint32_t)t;
}
__device__ __forceinline__ bool re_keep(uint32_t seed, uint32_t stream, uint32_t idx, uint32_t thresh) {
    return re_rng_u32(seed, stream, idx) >= thresh;
}
