// Owner bucketing of a batch's table lookups for the row-sharded table (row r lives on rank r mod G; "each GPU buckets
// its needed indices by owner -> all-to-all #1 -> owners gather -> all-to-all #2 -> un-permute").  One stable counting-sort pass
// over the G owners, no host sync and FIXED capacity per peer, so the exchange can use equal-split all-to-alls and the whole
// sharded step can be captured:
//   buckets [G][cap] int64 : the LOCAL row ids (r div G) wanted from owner g, in order of appearance, -1 in the unused slots
//   slot    [n]      int64 : where lookup j sits, g * cap + rank  (gather the received rows / scatter the gradient rows by it);
//                            -1 for a dropped lookup (index out of range, or the owner's bucket is full)
//   counts  [G + 1]  int32 : lookups per owner (may exceed cap), then the number of dropped lookups -- the caller checks [G] == 0
// skip_row (>= 0): lookups of that row -- the padding row, most of a left-padded batch -- take no bucket slot and are not counted as
// dropped; their slot is -1 (a zero row on the gathering side), so the capacity can be sized for the real tokens.
// Two launches: tile histograms (256 lookups per tile and wave-ballot ranks), then the placement (every workgroup derives its
// tile's bases from the [T][G] table itself -- the table is tiny).
#include "re_common.h"

#define RT_TILE 1024
#define RT_MAXG 64

// owner of row r; -1: out of range (a dropped lookup); -2: the skipped row
__device__ __forceinline__ int rt_owner(int64_t r, int64_t R, int G, int64_t skip) { return r == skip ? -2 : (r < 0 || r >= R) ? -1 : (int)(r % G); }

__global__ __launch_bounds__(256) void route_hist_k(const int64_t* __restrict__ idx, int64_t n, int64_t R, int G, int64_t skip, int* __restrict__ hist) {
    __shared__ int s_h[RT_MAXG];
    const int tid = threadIdx.x;
    if (tid < G) s_h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RT_TILE;
    for (int q = 0; q < RT_TILE / 256; ++q) {
        const int64_t j = base + q * 256 + tid;
        const int o = j < n ? rt_owner(idx[j], R, G, skip) : -1;
        for (int g = 0; g < G; ++g) {                       // (integer counts: order-free)
            const unsigned long long m = __ballot(o == g);
            if ((tid & 63) == 0 && m) atomicAdd(&s_h[g], __builtin_popcountll(m));
        }
    }
    __syncthreads();
    if (tid < G) hist[(int64_t)blockIdx.x * G + tid] = s_h[tid];
}

__global__ __launch_bounds__(256) void route_place_k(const int64_t* __restrict__ idx, int64_t n, int64_t R, int G, int64_t skip, int64_t cap, int ntiles,
                                                     const int* __restrict__ hist, int64_t* __restrict__ buckets, int64_t* __restrict__ slot,
                                                     int* __restrict__ counts) {
    __shared__ int s_base[RT_MAXG], s_run[RT_MAXG], s_w[4][RT_MAXG], s_drop;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // this tile's first rank per owner = the lookups of the tiles in front of it (threads over owners, a short loop over tiles)
    if (tid < G) {
        int b = 0;
        for (int t = 0; t < (int)blockIdx.x; ++t) b += hist[(int64_t)t * G + tid];
        s_base[tid] = b;
        s_run[tid] = 0;
        if (blockIdx.x == 0) {                              // totals (and the dropped count starts at the out-of-range lookups)
            int tot = b;
            for (int t = 0; t < ntiles; ++t) tot += hist[(int64_t)t * G + tid];
            counts[tid] = tot;
        }
    }
    if (tid == 0) s_drop = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RT_TILE;
    int dropped = 0;
    for (int q = 0; q < RT_TILE / 256; ++q) {
        const int64_t j = base + q * 256 + tid;
        const int64_t r = j < n ? idx[j] : -1;
        const int o = j < n ? rt_owner(r, R, G, skip) : -1;
        int rank = 0;
        for (int g = 0; g < G; ++g) {
            const unsigned long long m = __ballot(o == g);
            if (lane == 0) s_w[wave][g] = __builtin_popcountll(m);
            if (o == g) rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
        }
        __syncthreads();
        if (o >= 0) {
            for (int w = 0; w < wave; ++w) rank += s_w[w][o];
            rank += s_base[o] + s_run[o];
            if (rank < cap) {
                buckets[(int64_t)o * cap + rank] = r / G;
                slot[j] = (int64_t)o * cap + rank;
            } else {
                slot[j] = -1;
                ++dropped;
            }
        } else if (j < n) {
            slot[j] = -1;
            if (o == -1) ++dropped;
        }
        __syncthreads();
        if (tid < G) s_run[tid] += s_w[0][tid] + s_w[1][tid] + s_w[2][tid] + s_w[3][tid];
        __syncthreads();
    }
    if (dropped) atomicAdd(&s_drop, dropped);
    __syncthreads();
    if (tid == 0 && s_drop) atomicAdd(&counts[G], s_drop);
}

// unused bucket slots = -1, counts = 0 (before the placement)
__global__ __launch_bounds__(256) void route_fill_k(int64_t* __restrict__ buckets, int64_t total, int* __restrict__ counts, int G) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < total) buckets[i] = -1;
    if (i <= G) counts[i] = 0;
}

extern "C" size_t re_route_workspace_bytes(int64_t n, int64_t G) {
    if (n <= 0 || G <= 0) return 256;
    return re_align((size_t)re_cdiv(n, RT_TILE) * G * sizeof(int));
}

extern "C" int re_route_bucket(const int64_t* idx, int64_t n, int64_t R, int64_t G, int64_t cap, int64_t skip_row, int64_t* buckets,
                               int64_t* slot, int32_t* counts, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (G < 1 || G > RT_MAXG || cap < 1 || R < 1 || n < 0) return RE_EINVAL;
    if (!buckets || !counts || (n && (!idx || !slot || !ws))) return RE_EINVAL;
    if (n && ws_bytes < re_route_workspace_bytes(n, G)) return RE_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int ntiles = (int)re_cdiv(n, RT_TILE);
    hipLaunchKernelGGL(route_fill_k, dim3((unsigned)re_cdiv(G * cap, 256)), dim3(256), 0, s, buckets, G * cap, counts, (int)G);
    if (n == 0) {
        return re_launch_status();
    }
    hipLaunchKernelGGL(route_hist_k, dim3(ntiles), dim3(256), 0, s, idx, n, R, (int)G, skip_row, (int*)ws);
    hipLaunchKernelGGL(route_place_k, dim3(ntiles), dim3(256), 0, s, idx, n, R, (int)G, skip_row, cap, ntiles, (const int*)ws, buckets, slot, counts);
    return re_launch_status();
}
