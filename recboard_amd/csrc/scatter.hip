// K1b: dense embedding gradient  dW[r,:] = sum_{i: idx[i]==r} scale * g[i,:]   -- deterministic.
//
// Design (cdna_hip_programming.md Appendix B "Scatter / gather / embedding", the atomic-free form):
// the contribution rows already exist in HBM (`g`, the upstream gradient), so only an inverted index is built:
//   1. keys = destination row (R for dropped entries: padding_idx / out of range), vals = position i
//   2. LSD radix sort of (key, val), 8 bits per pass over ceil(log2(R+1)) bits; stable, so inside one
//      destination the positions stay ascending  -> a fixed summation order.
//      One wave per 1024-key tile; the in-tile stable rank uses wave64 ballots (8 per key byte) instead of
//      LDS atomics, per-digit running bases live in LDS.
//   3. segmented sum over fixed chunks of RE_SEG_CHUNK sorted entries: one lane group (D/4 lanes, float4 each)
//      per chunk walks its entries in order; runs that lie inside one chunk are written straight to dW, runs
//      that cross a chunk boundary leave a partial row; a fix-up kernel adds the partials in chunk order.
//      Hot rows (Zipf head: thousands of contributions) are thereby split across many lane groups
//      instead of serialising one wave (the skew pitfall of Appendix B).
// Float atomics would be ~1.3 TB/s of added bytes and run-to-run non-reproducible; this path reads each
// contribution row exactly once at gather-like rates and is bitwise reproducible.
//
// Algorithmic bytes: per contribution 8 (idx) + 4D (row read); per distinct row 4D (write).
#include <type_traits>

#include "re_common.h"

#define RE_SORT_TILE 1024
#define RE_SEG_CHUNK 16
#define RE_FLAG_SLOT0 1u      // chunk's first run continues a run of the previous chunk -> partial in slot 0
#define RE_FLAG_SLOT1 2u      // chunk's last run starts here and continues into the next chunk -> slot 1
#define RE_FLAG_CONT 4u       // slot-0 run covers the whole chunk and continues into the next one

// ------------------------------------------------------------------------------------------------ keys
__global__ __launch_bounds__(256) void scatter_make_keys(const int64_t* __restrict__ idx, int64_t n, int64_t R,
                                                         int64_t padding_idx, uint32_t* __restrict__ keys,
                                                         uint32_t* __restrict__ vals) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        int64_t r = idx[i];
        keys[i] = (r == padding_idx || r < 0 || r >= R) ? (uint32_t)R : (uint32_t)r;
        vals[i] = (uint32_t)i;
    }
}

// ------------------------------------------------------------------------------------------------ radix pass
__global__ __launch_bounds__(64) void radix_hist(const uint32_t* __restrict__ keys, int64_t n, int shift,
                                                 uint32_t* __restrict__ hist, int64_t T) {
    __shared__ uint32_t h[256];
    const int lane = threadIdx.x;
    for (int d = lane; d < 256; d += 64) h[d] = 0;
    __syncthreads();
    const int64_t tile = blockIdx.x;
    const int64_t base = tile * RE_SORT_TILE;
#pragma unroll 4
    for (int r = 0; r < RE_SORT_TILE / 64; ++r) {
        int64_t i = base + r * 64 + lane;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    for (int d = lane; d < 256; d += 64) hist[(int64_t)d * T + tile] = h[d];
}

// exclusive scan of hist[256*T] (digit-major, tile-minor) by ONE block of 1024 threads: each thread owns one
// contiguous chunk of ceil(total/1024) entries.  Chunks of <= 32 entries (n <= 131 072 keys) are held in registers:
// all loads are issued before the first use, so the kernel pays one memory latency instead of one per entry.
__global__ __launch_bounds__(1024) void radix_scan(uint32_t* __restrict__ hist, int64_t total) {
    __shared__ uint32_t wsum[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t chunk = (total + 1023) / 1024;
    const int64_t b = (int64_t)tid * chunk;
    const int64_t e = (b + chunk < total) ? b + chunk : total;
    const bool small = chunk <= 32;
    uint32_t v[32];
    uint32_t s = 0;
    if (small) {
#pragma unroll
        for (int i = 0; i < 32; ++i) v[i] = (b + i < e) ? hist[b + i] : 0u;
#pragma unroll
        for (int i = 0; i < 32; ++i) s += v[i];
    } else {
        for (int64_t i = b; i < e; ++i) s += hist[i];
    }
    uint32_t inc = s;  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wsum[wid] = inc;
    __syncthreads();
    uint32_t run = inc - s;
    for (int w = 0; w < wid; ++w) run += wsum[w];
    if (small) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (b + i < e) hist[b + i] = run;
            run += v[i];
        }
    } else {
        for (int64_t i = b; i < e; ++i) {
            const uint32_t x = hist[i];
            hist[i] = run;
            run += x;
        }
    }
}

__global__ __launch_bounds__(64) void radix_scatter(const uint32_t* __restrict__ keys_in,
                                                    const uint32_t* __restrict__ vals_in, int64_t n, int shift,
                                                    const uint32_t* __restrict__ hist, int64_t T,
                                                    uint32_t* __restrict__ keys_out,
                                                    uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t base[256];
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    for (int d = lane; d < 256; d += 64) base[d] = hist[(int64_t)d * T + tile];
    __syncthreads();
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    for (int r = 0; r < RE_SORT_TILE / 64; ++r) {
        const int64_t i = tile * RE_SORT_TILE + r * 64 + lane;
        const bool valid = i < n;
        uint32_t k = 0, v = 0;
        if (valid) { k = keys_in[i]; v = vals_in[i]; }
        const uint32_t d = (k >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & lt);
        uint32_t off = 0;
        if (valid) off = base[d] + rank;
        __syncthreads();
        if (valid && rank == 0) base[d] += __popcll(peers);
        __syncthreads();
        if (valid) { keys_out[off] = k; vals_out[off] = v; }
    }
}

// ------------------------------------------------------------------------------------------------ fused small-n sort
// n <= RE_FUSED_MAX_TILES * 1024 keys (a training batch: 76 800 contributions at SASRec/Beauty, 160 k at DeepFM/Criteo):
// the generic pipeline above is 1 + 3 launches per pass, each a latency-bound 64-thread tile walk.  Here a pass is two
// launches.  Tiles are 1024 keys on 256 threads (4 keys per lane, all loaded before the first ranking round); the scan
// kernel is folded into the scatter kernel (thread d sums column d of the [T x 256] tile histogram: everything below its
// tile, and the column total; one block scan over the totals gives the digit bases).  The first pass reads idx directly
// (no key/val arrays yet) and its histogram kernel also zero-fills dW.  (Building the next pass's histogram inside the
// scatter with global integer atomics was measured slower than a separate 256-thread histogram launch: 76 800
// contended atomics cost 14 us, the launch 4 us.)  Launches: 2 * passes (+ 2 for the segmented sum), vs 2 + 3 * passes (+ 2).
#define RE_FUSED_MAX_TILES 512

__device__ __forceinline__ uint32_t sc_key(int64_t r, int64_t R, int64_t padding_idx) {
    return (r == padding_idx || r < 0 || r >= R) ? (uint32_t)R : (uint32_t)r;
}

__global__ __launch_bounds__(256) void sc_hist0(const int64_t* __restrict__ idx, int64_t n, int64_t R, int64_t padding_idx,
                                                uint32_t* __restrict__ hist0, uint32_t* __restrict__ zero_u32, int64_t zero_words,
                                                float* __restrict__ zero_f32, int64_t zero_floats, int64_t zero_vec, int T) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    // zero fills that the later kernels rely on (independent of the histogram; blocks >= T exist only for this)
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < zero_words; i += (int64_t)gridDim.x * 256) zero_u32[i] = 0u;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < zero_vec; i += (int64_t)gridDim.x * 256) reinterpret_cast<float4*>(zero_f32)[i] = z;
    for (int64_t i = zero_vec * 4 + (int64_t)blockIdx.x * 256 + tid; i < zero_floats; i += (int64_t)gridDim.x * 256) zero_f32[i] = 0.f;
    if ((int)blockIdx.x >= T) return;
    h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RE_SORT_TILE;
    int64_t r[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + q * 256 + tid;
        r[q] = i < n ? idx[i] : -1;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (base + q * 256 + tid < n) atomicAdd(&h[sc_key(r[q], R, padding_idx) & 255u], 1u);
    __syncthreads();
    hist0[(int64_t)blockIdx.x * 256 + tid] = h[tid];
}

__global__ __launch_bounds__(256) void sc_histk(const uint32_t* __restrict__ keys, int64_t n, int shift, uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RE_SORT_TILE;
    uint32_t k[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int64_t i = base + q * 256 + tid;
        k[q] = i < n ? keys[i] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (base + q * 256 + tid < n) atomicAdd(&h[(k[q] >> shift) & 255u], 1u);
    __syncthreads();
    hist[(int64_t)blockIdx.x * 256 + tid] = h[tid];
}

// FIRST: keys come from idx (vals = position).
template <bool FIRST>
__global__ __launch_bounds__(256) void sc_scatter(const int64_t* __restrict__ idx, int64_t R, int64_t padding_idx,
                                                  const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, int64_t n,
                                                  int shift, const uint32_t* __restrict__ hist, int T,
                                                  uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t base[4][256];   // per wave: running output offset of each digit
    __shared__ uint32_t wcnt[4][256];   // per wave: digit counts of its 256-key strip
    __shared__ uint32_t wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    // ---- this block's digit bases: thread d owns digit d
    uint32_t below = 0, total = 0;
    {
        const uint32_t* col = hist + tid;
        int t = 0;
        for (; t + 8 <= T; t += 8) {
            uint32_t x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = col[(int64_t)(t + u) * 256];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                total += x[u];
                if (t + u < tile) below += x[u];
            }
        }
        for (; t < T; ++t) {
            const uint32_t x = col[(int64_t)t * 256];
            total += x;
            if (t < tile) below += x;
        }
    }
    uint32_t inc = total;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    if (lane == 63) wtot[wave] = inc;
    // ---- load this wave's strip (4 rounds of 64 keys) and count its digits
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
    uint32_t k[4], v[4];
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = (int64_t)tile * RE_SORT_TILE + wave * 256 + r * 64 + lane;
        ok[r] = i < n;
        k[r] = 0; v[r] = 0;
        if (ok[r]) {
            if (FIRST) { k[r] = sc_key(idx[i], R, padding_idx); v[r] = (uint32_t)i; }
            else { k[r] = keys_in[i]; v[r] = vals_in[i]; }
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (ok[r]) atomicAdd(&wcnt[wave][(k[r] >> shift) & 255u], 1u);
    __syncthreads();
    {
        uint32_t run = inc - total + below;
        for (int w = 0; w < wave; ++w) run += wtot[w];
        // run = first output slot of digit `tid` for this tile; split it over the 4 strips in order
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            base[w][tid] = run;
            run += wcnt[w][tid];
        }
    }
    __syncthreads();
    // ---- stable ranking, wave-private: 4 rounds of 64 keys
    const unsigned long long lt = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const uint32_t d = (k[r] >> shift) & 255u;
        unsigned long long peers = __ballot(ok[r]);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t rank = __popcll(peers & lt);
        uint32_t off = 0;
        if (ok[r]) off = base[wave][d] + rank;
        __syncthreads();
        if (ok[r] && rank == 0) base[wave][d] += __popcll(peers);
        __syncthreads();
        if (ok[r]) {
            keys_out[off] = k[r];
            vals_out[off] = v[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------ segmented sum
// column vectors: float4 (D % 4 == 0, 16-B aligned buffers) or float (any D: DeepFM's D = 10 and D = 1 tables)
__device__ __forceinline__ void f4_fma(float4& a, const float4& x, float s) {
    a.x = fmaf(x.x, s, a.x); a.y = fmaf(x.y, s, a.y); a.z = fmaf(x.z, s, a.z); a.w = fmaf(x.w, s, a.w);
}
__device__ __forceinline__ void f4_fma(float& a, const float& x, float s) { a = fmaf(x, s, a); }
__device__ __forceinline__ void f4_add(float4& a, const float4& x) { a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w; }
__device__ __forceinline__ void f4_add(float& a, const float& x) { a += x; }
template <class V> __device__ __forceinline__ V vzero();
template <> __device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float vzero<float>() { return 0.f; }

// What happens to a destination row's complete sum: written / added into the dense gradient table, or -- row-sparse
// optimizer for tables too large for a dense gradient -- fed straight into an Adam update of that row.
struct DenseSink {
    float* dW;
    int64_t D;
    int accumulate;
    template <class V>
    __device__ __forceinline__ void put(uint32_t row, int64_t col, V acc) const {
        V* d = reinterpret_cast<V*>(dW + (int64_t)row * D) + col;
        if (accumulate) f4_add(acc, *d);
        *d = acc;
    }
};
struct AdamSink {   // torch.optim.SparseAdam semantics on the touched rows (+ coupled weight decay on them), global step count
    float *W, *m, *v;
    int64_t D;
    float b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd;
    const float* hyper;   // non-null: { lr / (1 - b1^t), 1 / sqrt(1 - b2^t) } in device memory (hipGraph replays)
    __device__ __forceinline__ void one(float& p, float& mm, float& vv, float g) const {
        const float ss = hyper ? hyper[0] : step_size, ib = hyper ? hyper[1] : inv_sqrt_bc2;
        const float gg = g + wd * p;
        mm = b1 * mm + omb1 * gg;
        vv = b2 * vv + omb2 * gg * gg;
        p = p - ss * (mm / (sqrtf(vv) * ib + eps));
    }
    __device__ __forceinline__ void put(uint32_t row, int64_t col, float4 acc) const {
        float4* pp = reinterpret_cast<float4*>(W + (int64_t)row * D) + col;
        float4* pm = reinterpret_cast<float4*>(m + (int64_t)row * D) + col;
        float4* pv = reinterpret_cast<float4*>(v + (int64_t)row * D) + col;
        float4 P = *pp, M = *pm, Vv = *pv;
        one(P.x, M.x, Vv.x, acc.x); one(P.y, M.y, Vv.y, acc.y); one(P.z, M.z, Vv.z, acc.z); one(P.w, M.w, Vv.w, acc.w);
        *pp = P; *pm = M; *pv = Vv;
    }
    __device__ __forceinline__ void put(uint32_t row, int64_t col, float acc) const {
        const int64_t o = (int64_t)row * D + col;
        float P = W[o], M = m[o], Vv = v[o];
        one(P, M, Vv, acc);
        W[o] = P; m[o] = M; v[o] = Vv;
    }
};

template <int LPR, class V, class Sink>
__global__ __launch_bounds__(256) void seg_reduce(const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals,
                                                  const float* __restrict__ g, int64_t n, int64_t D, int64_t R,
                                                  float scale, Sink sink, float* __restrict__ partial,
                                                  uint32_t* __restrict__ pflags, int64_t nchunks) {
    const int lir = threadIdx.x % LPR;
    const int64_t c = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (c >= nchunks) return;
    const int64_t begin = c * RE_SEG_CHUNK;
    const int64_t end = (begin + RE_SEG_CHUNK < n) ? begin + RE_SEG_CHUNK : n;
    const uint32_t NONE = 0xFFFFFFFFu, DROP = (uint32_t)R;
    const uint32_t prevKey = begin > 0 ? keys[begin - 1] : NONE;
    const uint32_t nextKey = end < n ? keys[end] : NONE;
    const int64_t D4 = D / (int64_t)(sizeof(V) / sizeof(float));
    uint32_t flags = 0;
    // Dropped entries (padding / out of range) sort to the end: a chunk that STARTS with one consists of nothing else, and
    // its rows are not even read -- in a SASRec batch 88 % of the 3*B*S contribution rows are padding positions.
    if (keys[begin] == DROP) {
        if (lir == 0) pflags[c] = 0;
        return;
    }
    for (int64_t col = lir; col < D4; col += LPR) {
        uint32_t curKey = keys[begin];
        bool isHead = true;
        V acc = vzero<V>();
        for (int64_t j0 = begin; j0 < end; j0 += RE_SEG_CHUNK) {   // (one trip: every load of the chunk is in flight at once)
            uint32_t k[RE_SEG_CHUNK];
            V row[RE_SEG_CHUNK];
            // two dependent load stages, no branches in between: (keys, positions) of the whole chunk, then every row --
            // a dropped entry's position is still a valid row of g, so its row is simply loaded and not used
            uint32_t pos[RE_SEG_CHUNK];
#pragma unroll
            for (int u = 0; u < RE_SEG_CHUNK; ++u) {
                const int64_t j = (j0 + u < end) ? j0 + u : end - 1;
                k[u] = keys[j];
                pos[u] = vals[j];
            }
#pragma unroll
            for (int u = 0; u < RE_SEG_CHUNK; ++u) row[u] = reinterpret_cast<const V*>(g + (int64_t)pos[u] * D)[col];
#pragma unroll
            for (int u = 0; u < RE_SEG_CHUNK; ++u) {
                if (j0 + u >= end) k[u] = NONE;
                if (k[u] == DROP) row[u] = vzero<V>();
            }
#pragma unroll
            for (int u = 0; u < RE_SEG_CHUNK; ++u) {
                if (j0 + u >= end) break;
                if (k[u] != curKey) {
                    // flush a run that ended strictly inside the chunk (cannot be the tail)
                    if (curKey != DROP) {
                        if (isHead && curKey == prevKey) {
                            reinterpret_cast<V*>(partial + (c * 2 + 0) * D)[col] = acc;
                            flags |= RE_FLAG_SLOT0;
                        } else {
                            sink.put(curKey, col, acc);
                        }
                    }
                    curKey = k[u];
                    isHead = false;
                    acc = vzero<V>();
                }
                f4_fma(acc, row[u], scale);
            }
        }
        // flush the tail run
        if (curKey != DROP) {
            const bool headOpen = isHead && curKey == prevKey;
            const bool tailOpen = curKey == nextKey;
            if (headOpen) {
                reinterpret_cast<V*>(partial + (c * 2 + 0) * D)[col] = acc;
                flags |= RE_FLAG_SLOT0 | (tailOpen ? RE_FLAG_CONT : 0u);
            } else if (tailOpen) {
                reinterpret_cast<V*>(partial + (c * 2 + 1) * D)[col] = acc;
                flags |= RE_FLAG_SLOT1;
            } else {
                sink.put(curKey, col, acc);
            }
        }
    }
    if (lir == 0) pflags[c] = flags;
}

// one lane group per chunk in which a boundary-crossing run STARTS: add the following chunks' slot-0 partials in order.
// Chains of up to 8 links (the common case) are finished by their lane group alone.  A longer chain -- a hot row: a Zipf-head
// item, a low-cardinality DeepFM field -- is handed to the whole block: its links are split into G = 256/LPR contiguous
// ranges summed concurrently (each in link order), and the G range sums are added in range order.  The summation tree is a
// fixed function of the sorted keys, so the result stays bitwise reproducible.
template <int LPR, class V, class Sink>
__global__ __launch_bounds__(256) void seg_fixup(const uint32_t* __restrict__ keys, int64_t n, int64_t D,
                                                 Sink sink, const float* __restrict__ partial,
                                                 const uint32_t* __restrict__ pflags, int64_t nchunks) {
    constexpr int G = 256 / LPR;
    __shared__ int64_t long_c[G];
    __shared__ int n_long;
    __shared__ unsigned long long chain_end;
    __shared__ V red[G][LPR];
    const int lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const int64_t D4 = D / (int64_t)(sizeof(V) / sizeof(float));
    if (threadIdx.x == 0) n_long = 0;
    __syncthreads();
    const int64_t c = (int64_t)blockIdx.x * G + grp;
    if (c < nchunks && (pflags[c] & RE_FLAG_SLOT1)) {
        const int64_t end = (c * RE_SEG_CHUNK + RE_SEG_CHUNK < n) ? c * RE_SEG_CHUNK + RE_SEG_CHUNK : n;
        const uint32_t key = keys[end - 1];
        bool is_long = false;
        for (int64_t col = lir; col < D4 && !is_long; col += LPR) {
            V acc = reinterpret_cast<const V*>(partial + (c * 2 + 1) * D)[col];
            bool open = true;
            uint32_t f[8];
            V part[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {   // 8 links of the chain, loads issued together
                const int64_t cc = (c + 1 + u < nchunks) ? c + 1 + u : nchunks - 1;
                f[u] = (c + 1 + u < nchunks) ? pflags[cc] : 0u;
                part[u] = reinterpret_cast<const V*>(partial + (cc * 2 + 0) * D)[col];   // (valid memory; used only on the chain)
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (!open) break;
                if (!(f[u] & RE_FLAG_SLOT0)) { open = false; break; }  // cannot happen for a well-formed chain
                f4_add(acc, part[u]);
                if (!(f[u] & RE_FLAG_CONT)) open = false;
            }
            if (open && c + 9 < nchunks) { is_long = true; break; }   // (uniform over the lane group: depends on flags only)
            sink.put(key, col, acc);
        }
        if (is_long && lir == 0) long_c[atomicAdd(&n_long, 1)] = c;
    }
    __syncthreads();
    const int nl = n_long;
    for (int l = 0; l < nl; ++l) {
        // deterministic pick: the l-th smallest registered chunk (registration order is not deterministic)
        int64_t lc = 0;
        {
            int64_t prev = -1;
            for (int q = 0; q <= l; ++q) {
                int64_t best = INT64_MAX;
                for (int i = 0; i < nl; ++i)
                    if (long_c[i] > prev && long_c[i] < best) best = long_c[i];
                prev = best;
            }
            lc = prev;
        }
        // ---- where the chain ends: the first link without CONT
        if (threadIdx.x == 0) chain_end = ~0ull;
        __syncthreads();
        for (int64_t j0 = lc + 1; j0 < nchunks; j0 += 256) {
            const int64_t cc = j0 + threadIdx.x;
            if (cc < nchunks) {
                const uint32_t f = pflags[cc];
                if (!(f & RE_FLAG_CONT) || !(f & RE_FLAG_SLOT0)) atomicMin(&chain_end, (unsigned long long)cc);
            }
            __syncthreads();
            if (chain_end != ~0ull) break;
            __syncthreads();
        }
        __syncthreads();
        int64_t last = (chain_end == ~0ull) ? nchunks - 1 : (int64_t)chain_end;   // last link (inclusive)
        if (!(pflags[last] & RE_FLAG_SLOT0)) --last;                               // malformed chain guard
        const int64_t m = last - lc;                                               // links lc+1 .. last
        const int64_t per = (m + G - 1) / G;
        const int64_t lo = lc + 1 + (int64_t)grp * per;
        const int64_t hi = (lo + per < last + 1) ? lo + per : last + 1;
        const int64_t end = (lc * RE_SEG_CHUNK + RE_SEG_CHUNK < n) ? lc * RE_SEG_CHUNK + RE_SEG_CHUNK : n;
        const uint32_t key = keys[end - 1];
        for (int64_t col0 = 0; col0 < D4; col0 += LPR) {
            const int64_t col = col0 + lir;
            V acc = vzero<V>();
            if (col < D4) {
                for (int64_t c0 = lo; c0 < hi; c0 += 8) {
                    V part[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int64_t cc = (c0 + u < hi) ? c0 + u : hi - 1;
                        part[u] = reinterpret_cast<const V*>(partial + (cc * 2 + 0) * D)[col];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (c0 + u < hi) f4_add(acc, part[u]);
                }
            }
            red[grp][lir] = acc;
            __syncthreads();
            if (grp == 0 && col < D4) {
                V tot = reinterpret_cast<const V*>(partial + (lc * 2 + 1) * D)[col];
                for (int q = 0; q < G; ++q) f4_add(tot, red[q][lir]);
                sink.put(key, col, tot);
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct ScatterWs {
    uint32_t *k0, *k1, *v0, *v1, *hist, *pflags;
    float* partial;
    int64_t T, nchunks;
    size_t bytes;
};

static ScatterWs scatter_ws_layout(void* ws, int64_t n, int64_t D) {
    ScatterWs w;
    w.T = re_cdiv(n, RE_SORT_TILE);
    w.nchunks = re_cdiv(n, RE_SEG_CHUNK);
    size_t off = 0;
    char* base = (char*)ws;
    auto take = [&](size_t bytes) { char* p = base + off; off += re_align(bytes); return p; };
    w.k0 = (uint32_t*)take((size_t)n * 4);
    w.k1 = (uint32_t*)take((size_t)n * 4);
    w.v0 = (uint32_t*)take((size_t)n * 4);
    w.v1 = (uint32_t*)take((size_t)n * 4);
    w.hist = (uint32_t*)take((size_t)256 * w.T * 4 * 4);   // up to 4 passes of [T x 256] (fused path keeps one per pass)
    w.pflags = (uint32_t*)take((size_t)w.nchunks * 4);
    w.partial = (float*)take((size_t)w.nchunks * 2 * D * 4);
    w.bytes = off;
    return w;
}

extern "C" size_t re_scatter_add_rows_workspace_bytes(int64_t n, int64_t D, int64_t R) {
    (void)R;
    if (n <= 0 || D <= 0) return 256;
    return scatter_ws_layout(nullptr, n, D).bytes;
}

// ---- the index half (sort) and the data half (segmented sum) are separate entry points: the sort depends on idx only,
//      so a training step can run it on a second stream while the gradients are still being computed.
static int scatter_passes(int64_t R) {
    int bits = 1;
    while (((int64_t)1 << bits) <= R) ++bits;  // keys take values 0..R
    return (bits + 7) / 8;
}

static int scatter_sort(const int64_t* idx, int64_t n, int64_t R, int64_t padding_idx, float* zero_fill, int64_t zfloats,
                        const ScatterWs& w, hipStream_t s) {
    const int passes = scatter_passes(R);
    uint32_t *ki = w.k0, *vi = w.v0, *ko = w.k1, *vo = w.v1;
    if (w.T <= RE_FUSED_MAX_TILES) {
        const int T = (int)w.T;
        int64_t zblocks = re_cdiv(zfloats, 4096);   // >= 16 KB of zero fill per block
        if (zblocks > 2048) zblocks = 2048;
        hipLaunchKernelGGL(sc_hist0, dim3((unsigned)(zblocks > T ? zblocks : T)), dim3(256), 0, s, idx, n, R, padding_idx, w.hist,
                           (uint32_t*)nullptr, (int64_t)0, zero_fill, zfloats,
                           (reinterpret_cast<uintptr_t>(zero_fill) & 15u) ? (int64_t)0 : zfloats >> 2, T);
        for (int p = 0; p < passes; ++p) {
            if (p == 0) {
                hipLaunchKernelGGL(sc_scatter<true>, dim3((unsigned)T), dim3(256), 0, s, idx, R, padding_idx, (const uint32_t*)nullptr,
                                   (const uint32_t*)nullptr, n, 0, w.hist, T, ko, vo);
            } else {
                hipLaunchKernelGGL(sc_histk, dim3((unsigned)T), dim3(256), 0, s, ki, n, 8 * p, w.hist);
                hipLaunchKernelGGL(sc_scatter<false>, dim3((unsigned)T), dim3(256), 0, s, idx, R, padding_idx, ki, vi, n, 8 * p, w.hist, T, ko, vo);
            }
            uint32_t* t;
            t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    } else {
        if (zfloats > 0 && re_zero_async(zero_fill, (size_t)zfloats * sizeof(float), s) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(scatter_make_keys, dim3(re_grid(n, 256)), dim3(256), 0, s, idx, n, R, padding_idx, w.k0, w.v0);
        for (int p = 0; p < passes; ++p) {
            hipLaunchKernelGGL(radix_hist, dim3((unsigned)w.T), dim3(64), 0, s, ki, n, 8 * p, w.hist, w.T);
            hipLaunchKernelGGL(radix_scan, dim3(1), dim3(1024), 0, s, w.hist, (int64_t)256 * w.T);
            hipLaunchKernelGGL(radix_scatter, dim3((unsigned)w.T), dim3(64), 0, s, ki, vi, n, 8 * p, w.hist, w.T, ko, vo);
            uint32_t* t;
            t = ki; ki = ko; ko = t;
            t = vi; vi = vo; vo = t;
        }
    }
    return RE_OK;
}

template <class Sink>
static void scatter_reduce_sink(const float* g, int64_t n, int64_t D, int64_t R, float scale, const Sink& sink, bool vec, const ScatterWs& w,
                                hipStream_t s) {
    const bool odd = scatter_passes(R) & 1;   // every pass ping-pongs the key / value arrays
    const uint32_t* ki = odd ? w.k1 : w.k0;
    const uint32_t* vi = odd ? w.v1 : w.v0;
#define SEG_LAUNCH(LPRV, VT)                                                                                                       \
    do {                                                                                                                           \
        const unsigned grid = (unsigned)re_cdiv(w.nchunks, 256 / LPRV);                                                            \
        hipLaunchKernelGGL((seg_reduce<LPRV, VT, Sink>), dim3(grid), dim3(256), 0, s, ki, vi, g, n, D, R, scale, sink, w.partial, w.pflags, w.nchunks); \
        hipLaunchKernelGGL((seg_fixup<LPRV, VT, Sink>), dim3(grid), dim3(256), 0, s, ki, n, D, sink, w.partial, w.pflags, w.nchunks); \
    } while (0)
    if (vec) {
        const int64_t D4 = D >> 2;
        if (D4 >= 32) SEG_LAUNCH(32, float4);
        else if (D4 >= 16) SEG_LAUNCH(16, float4);
        else SEG_LAUNCH(4, float4);
    } else {  // one lane per column (DeepFM: D = 10 -> 16 lanes, 10 active; D = 1 -> 4 lanes), same chunked algorithm
        // (D <= 4 too on 16-lane groups: with 4-lane groups a workgroup owns 64 chunks, and the hot rows of a low-cardinality field -- chains
        //  of > 8 chunks, finished one after the other by the whole workgroup -- piled up: 29 us for DeepFM's [40 960, 1] LR gradient against
        //  7 us for its [40 960, 10] rows on the same keys)
        if (D > 16) SEG_LAUNCH(32, float);
        else SEG_LAUNCH(16, float);
    }
#undef SEG_LAUNCH
}

static void scatter_reduce(const float* g, int64_t n, int64_t D, int64_t R, float scale, float* dW, int accumulate, const ScatterWs& w,
                           hipStream_t s) {
    const bool vec = (D & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dW)) & 15u) == 0;
    scatter_reduce_sink(g, n, D, R, scale, DenseSink{dW, D, accumulate}, vec, w, s);
}

extern "C" int re_scatter_plan(const int64_t* idx, int64_t n, int64_t D, int64_t R, int64_t padding_idx, float* zero_fill,
                               int64_t zero_floats, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (R <= 0 || D <= 0 || n < 0 || zero_floats < 0 || (zero_floats > 0 && !zero_fill)) return RE_EINVAL;
    if (R >= 0xFFFFFFFEll || n >= 0xFFFFFFFFll) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (zero_floats > 0 && re_zero_async(zero_fill, (size_t)zero_floats * sizeof(float), s) != hipSuccess) return RE_ELAUNCH;
        return RE_OK;
    }
    if (!idx || !ws) return RE_EINVAL;
    ScatterWs w = scatter_ws_layout(ws, n, D);
    if (ws_bytes < w.bytes) return RE_EWORKSPACE;
    const int rc = scatter_sort(idx, n, R, padding_idx, zero_fill, zero_floats, w, s);
    return rc != RE_OK ? rc : re_launch_status();
}

extern "C" int re_scatter_apply(const float* g, int64_t n, int64_t D, int64_t R, float scale, float* dW, int accumulate, void* ws,
                                size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!dW || R <= 0 || D <= 0 || n < 0) return RE_EINVAL;
    if (R >= 0xFFFFFFFEll || n >= 0xFFFFFFFFll) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (accumulate < 0 || accumulate > 2) return RE_EINVAL;
    if (n == 0) {
        if (!accumulate && re_zero_async(dW, (size_t)R * D * sizeof(float), s) != hipSuccess) return RE_ELAUNCH;
        return RE_OK;
    }
    if (!g || !ws) return RE_EINVAL;
    ScatterWs w = scatter_ws_layout(ws, n, D);
    if (ws_bytes < w.bytes) return RE_EWORKSPACE;
    // accumulate: 0 = dW is the scatter's dense result (zero-filled first); 1 = added to dW; 2 = ONLY the rows that occur are written (assigned),
    // the others keep what they hold -- for a consumer that knows which rows those are (re_row_mask + re_spmm_csr_masked: a 31 MB fill less)
    if (accumulate == 2) { scatter_reduce(g, n, D, R, scale, dW, 0, w, s); return re_launch_status(); }
    if (!accumulate && re_zero_async(dW, (size_t)R * D * sizeof(float), s) != hipSuccess) return RE_ELAUNCH;
    scatter_reduce(g, n, D, R, scale, dW, 1, w, s);
    return re_launch_status();
}

extern "C" int re_scatter_add_rows(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R,
                                   int64_t padding_idx, float scale, float* dW, int accumulate, void* ws, size_t ws_bytes,
                                   re_stream_t stream) {
    re_clear_error();
    if (!dW || R <= 0 || D <= 0 || n < 0) return RE_EINVAL;
    if (R >= 0xFFFFFFFEll || n >= 0xFFFFFFFFll) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (!accumulate && re_zero_async(dW, (size_t)R * D * sizeof(float), s) != hipSuccess) return RE_ELAUNCH;
        return RE_OK;
    }
    if (!g || !idx || !ws) return RE_EINVAL;
    ScatterWs w = scatter_ws_layout(ws, n, D);
    if (ws_bytes < w.bytes) return RE_EWORKSPACE;
    const int rc = scatter_sort(idx, n, R, padding_idx, dW, accumulate ? (int64_t)0 : (int64_t)R * D, w, s);
    if (rc != RE_OK) return rc;
    scatter_reduce(g, n, D, R, scale, dW, accumulate, w, s);
    return re_launch_status();
}

// Row-sparse Adam: for every distinct destination row r of idx (padding / out-of-range entries dropped), G_r = sum of the rows of g
// that point at it (position order, deterministic), then one Adam update of row r of (W, m, v) with the GLOBAL step count --
// torch.optim.SparseAdam's rule plus coupled weight decay on the touched rows.  Rows that receive no gradient are not touched
// (a dense Adam would keep decaying their moments): the optimizer for tables whose dense gradient does not fit (config 5).
static int sparse_adam_launch(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R, int64_t padding_idx, float* W, float* m,
                              float* v, float step_size, float inv_sqrt_bc2, const float* hyper, double beta1, double beta2, double eps,
                              double weight_decay, void* ws, size_t ws_bytes, re_stream_t stream) {
    if (!W || !m || !v || R <= 0 || D <= 0 || n < 0) return RE_EINVAL;
    if (R >= 0xFFFFFFFEll || n >= 0xFFFFFFFFll) return RE_EUNSUPPORTED;
    if (n == 0) return RE_OK;
    if (!g || !idx || !ws) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    ScatterWs w = scatter_ws_layout(ws, n, D);
    if (ws_bytes < w.bytes) return RE_EWORKSPACE;
    const int rc = scatter_sort(idx, n, R, padding_idx, nullptr, 0, w, s);
    if (rc != RE_OK) return rc;
    AdamSink sink{W, m, v, D, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), step_size, inv_sqrt_bc2, (float)eps,
                  (float)weight_decay, hyper};
    const bool vec = (D & 3) == 0 && ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(m) |
                                       reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    scatter_reduce_sink(g, n, D, R, 1.0f, sink, vec, w, s);
    return re_launch_status();
}

extern "C" int re_sparse_adam_rows(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R, int64_t padding_idx, float* W,
                                   float* m, float* v, int64_t step, double lr, double beta1, double beta2, double eps,
                                   double weight_decay, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (step < 1) return RE_EINVAL;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    return sparse_adam_launch(g, idx, n, D, R, padding_idx, W, m, v, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), nullptr, beta1, beta2, eps,
                              weight_decay, ws, ws_bytes, stream);
}

// hipGraph-friendly form: the two step-dependent scalars come from device memory (as re_adam_step_dev; written by re_sasrec_batch_prep)
extern "C" int re_sparse_adam_rows_dev(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R, int64_t padding_idx, float* W,
                                       float* m, float* v, const float* hyper, double beta1, double beta2, double eps,
                                       double weight_decay, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!hyper) return RE_EINVAL;
    return sparse_adam_launch(g, idx, n, D, R, padding_idx, W, m, v, 0.f, 0.f, hyper, beta1, beta2, eps, weight_decay, ws, ws_bytes, stream);
}

// ------------------------------------------------------------------------------------------------ small dense tables: owner computes
// (the algorithm: scatter_owner.h)
#include "scatter_owner.h"

template <int D, int HS, int NG = SO_NG>
__global__ __launch_bounds__(SO_NT) void scatter_owner_k(const float* __restrict__ g, const int32_t* __restrict__ keys, int nreg, int64_t stride,
                                                         const int32_t* __restrict__ n_dev, int n_mul, int64_t n_host, int64_t R, int rpw,
                                                         int64_t padding_idx, float scale, float* __restrict__ dW, SoAdam AD) {
    extern __shared__ __align__(16) float so_acc[];
    so_body<D, HS, SoNoHook, NG>(g, keys, nreg, stride, n_dev, n_mul, n_host, R, rpw, padding_idx, scale, dW, AD, so_acc);
}

static int scatter_small_launch(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev,
                                int32_t n_mul, int64_t n_host, int64_t D, int64_t R, int64_t padding_idx, float scale, float* dW, const SoAdam& AD,
                                re_stream_t stream) {
    re_clear_error();
    if ((!dW && !AD.W) || !g || !keys || R <= 0 || n_regions < 1 || n_regions > 4 || region_stride < 0 || n_host < 0 || (n_dev && n_mul < 1)) return RE_EINVAL;
    if (D != 64 && D != 128) return RE_EUNSUPPORTED;
    if ((region_stride & 3) || ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dW) |
                                 reinterpret_cast<uintptr_t>(AD.W) | reinterpret_cast<uintptr_t>(AD.m) | reinterpret_cast<uintptr_t>(AD.v)) & 15u))
        return RE_EUNSUPPORTED;
    if ((int64_t)n_regions * region_stride >= (1ll << 25)) return RE_EUNSUPPORTED;   // (a list entry: 7 bits of local row, 25 of index)
    constexpr int HS = 2;                          // column pieces per row
    int rpw = D == 64 ? 96 : 48;                   // virtual rows per workgroup: 96 KB of accumulators (8 sets x rpw x D / HS floats)
    int64_t nwg = HS;
    while (nwg * rpw < R * HS) nwg *= 2;          // a power of two: virtual rows are dealt round-robin with a mask
    // More rows than 256 workgroups hold at eight accumulator sets: two sets and four times the rows per workgroup (a workgroup is 144 KB of
    // LDS: one per CU, so every further 256 workgroups are another pass over the chip) -- where a list entry's 23 index bits suffice.
    const bool wide = nwg > 256 && (int64_t)n_regions * region_stride < (1ll << 23);
    if (wide) {
        rpw *= 4;
        nwg = 256;
        while (nwg * rpw < R * HS) nwg *= 2;
    }
    if (nwg > 4096) return RE_EUNSUPPORTED;       // every workgroup scans all keys: past ~100 k rows the sorted path is the right one
    const size_t ldsb = (size_t)(wide ? 2 : SO_NG) * rpw * (D / HS) * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define SO_LAUNCH(DV, NGV)                                                                                                                       \
    do {                                                                                                                                          \
        auto k = scatter_owner_k<DV, HS, NGV>;                                                                                                    \
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;          \
        hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(SO_NT), ldsb, s, g, keys, (int)n_regions, region_stride, n_dev, (int)n_mul, n_host, R, rpw, \
                           padding_idx, scale, dW, AD);                                                                                           \
    } while (0)
    if (D == 64) { if (wide) SO_LAUNCH(64, 2); else SO_LAUNCH(64, SO_NG); }
    else { if (wide) SO_LAUNCH(128, 2); else SO_LAUNCH(128, SO_NG); }
#undef SO_LAUNCH
    return re_launch_status();
}

extern "C" int re_scatter_add_rows_small(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev,
                                         int32_t n_mul, int64_t n_host, int64_t D, int64_t R, int64_t padding_idx, float scale, float* dW,
                                         re_stream_t stream) {
    if (!dW) return RE_EINVAL;
    SoAdam AD{};
    return scatter_small_launch(g, keys, n_regions, region_stride, n_dev, n_mul, n_host, D, R, padding_idx, scale, dW, AD, stream);
}

extern "C" int re_scatter_adam_rows_small(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev,
                                          int32_t n_mul, int64_t n_host, int64_t D, int64_t R, int64_t padding_idx, float scale, float* dW,
                                          const re_adam_fuse* adam, re_stream_t stream) {
    if (!adam || !adam->param || !adam->m || !adam->v || !adam->hyper) return RE_EINVAL;
    const SoAdam AD{adam->param, adam->m, adam->v, adam->hyper, (float)adam->beta1, (float)adam->beta2, (float)(1.0 - adam->beta1),
                    (float)(1.0 - adam->beta2), (float)adam->eps, (float)adam->weight_decay};
    return scatter_small_launch(g, keys, n_regions, region_stride, n_dev, n_mul, n_host, D, R, padding_idx, scale, dW, AD, stream);
}
