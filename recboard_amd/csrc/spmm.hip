// K8: CSR SpMM for LightGCN propagation:  Y = A X (+ beta Z),  optionally  ACC += acc_scale * Y.      HBM / Infinity-Cache bound.
//
// Reference: `allEmbds = self.Adj @ allEmbds; avgEmbds += allEmbds / (L+1)` (LightGCN/main.py:81-84), Adj = the
// symmetric-normalised bipartite adjacency as a CSR tensor (LightGCN/main.py:47-49); aten runs rocSPARSE csrmm plus an
// elementwise add per layer, and the transposed product in backward.  Adj is symmetric, so backward is the same kernel:
//   g_l = A g_{l+1} + dAvg / (L+1)        (Z = dAvg, beta = 1/(L+1))
//
// Layout: one lane group (D/4 lanes x float4) per output row walks the row's non-zeros four at a time (independent
// col/val loads, then four independent X-row loads); a row's 256-B X rows are full-line reads.  X (31.5 MB at Yelp
// sizes) lives in the 256 MB Infinity Cache, so the stream that must come from HBM is (col, val) = 12 B per non-zero.
// Power-law rows: the caller passes the rows in descending-degree order (computed once per adjacency); the first
// `nlong` of them (more than SP_LONG non-zeros) get a whole workgroup each (16 lane groups, strided non-zeros,
// fixed-order LDS combine), the rest are walked in that order so the rows sharing a wave have similar lengths -- the
// per-wave skew pitfall of cdna_hip_programming.md Appendix B.  Summation order is fixed => bitwise reproducible.
//
// Algorithmic bytes: per non-zero 12 B (+ a 4D-byte X row, cache-resident); per row 8 B crow + 4D B write
// (+ 4D B for Z, + 8D B for ACC); 2D FLOP per non-zero.
#include "re_common.h"

#define SP_LONG 512

__device__ __forceinline__ void f4_axpy(float4& a, float s, const float4& x) {
    a.x = fmaf(s, x.x, a.x); a.y = fmaf(s, x.y, a.y); a.z = fmaf(s, x.z, a.z); a.w = fmaf(s, x.w, a.w);
}

#ifndef SP_ILP
#define SP_ILP 8
#endif
// a row's (or a strided share of a row's) non-zeros SP_ILP at a time; the NEXT group's (col, val) are requested before this group's X rows, so
// an iteration costs one memory round trip (the X rows), not two (indices, then rows)
template <int LPR>
__device__ __forceinline__ float4 spmm_row_range(const int64_t* __restrict__ col, const float* __restrict__ val,
                                                  const float* __restrict__ X, int64_t ncols, int64_t D, int64_t c4,
                                                  int64_t p0, int64_t p1, int64_t step) {
    constexpr int U = SP_ILP;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t p = p0;
    if (p + (U - 1) * step < p1) {
        int64_t cc[U];
        float vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { cc[u] = col[p + u * step]; vv[u] = val[p + u * step]; }
        for (;;) {
            const int64_t pn = p + U * step;
            const bool more = pn + (U - 1) * step < p1;
            int64_t nc[U];
            float nv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {   // (clamped, unconditional: in flight beside the X rows below.  Padding the last group with
                const int64_t q = more ? pn + u * step : p;   //  weight-0 slots instead of the tail loops was measured slower: 136 / 156 vs 126 us)
                nc[u] = col[q]; nv[u] = val[q];
            }
            float4 xr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cc[u] >= 0 && cc[u] < ncols) xr[u] = reinterpret_cast<const float4*>(X + cc[u] * D)[c4];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) f4_axpy(acc, vv[u], xr[u]);
            p = pn;
            if (!more) break;
#pragma unroll
            for (int u = 0; u < U; ++u) { cc[u] = nc[u]; vv[u] = nv[u]; }
        }
    }
    // the tail: up to U - 1 non-zeros, four at a time, then one by one
    for (; p + 3 * step < p1; p += 4 * step) {
        int64_t cc[4];
        float vv[4];
        float4 xr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { cc[u] = col[p + u * step]; vv[u] = val[p + u * step]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cc[u] >= 0 && cc[u] < ncols) xr[u] = reinterpret_cast<const float4*>(X + cc[u] * D)[c4];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) f4_axpy(acc, vv[u], xr[u]);
    }
    for (; p < p1; p += step) {
        const int64_t cc = col[p];
        if (cc >= 0 && cc < ncols) f4_axpy(acc, val[p], reinterpret_cast<const float4*>(X + cc * D)[c4]);
    }
    return acc;
}

template <int LPR>
__device__ __forceinline__ void spmm_store(float4 acc, int64_t r, int64_t D, int64_t c4, float* __restrict__ Y,
                                           const float* __restrict__ Z, float beta, float* __restrict__ ACC, float acc_scale) {
    if (Z) f4_axpy(acc, beta, reinterpret_cast<const float4*>(Z + r * D)[c4]);
    reinterpret_cast<float4*>(Y + r * D)[c4] = acc;
    if (ACC) {
        float4 a = reinterpret_cast<float4*>(ACC + r * D)[c4];
        f4_axpy(a, acc_scale, acc);
        reinterpret_cast<float4*>(ACC + r * D)[c4] = a;
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void spmm_csr_rows(const int64_t* __restrict__ row_order, int64_t first,
                                                     const int64_t* __restrict__ crow, const int64_t* __restrict__ col,
                                                     const float* __restrict__ val, int64_t nrows, int64_t ncols,
                                                     const float* __restrict__ X, int64_t D, float* __restrict__ Y,
                                                     const float* __restrict__ Z, float beta, float* __restrict__ ACC,
                                                     float acc_scale) {
    const int lir = threadIdx.x % LPR;
    const int64_t gpb = 256 / LPR;
    const int64_t D4 = D >> 2;
    // rows are visited in descending-degree order (row_order): the 4 (or 2) rows a wave works on have similar lengths,
    // and the longest rows start first
    for (int64_t i = first + (int64_t)blockIdx.x * gpb + threadIdx.x / LPR; i < nrows; i += (int64_t)gridDim.x * gpb) {
        const int64_t r = row_order ? row_order[i] : i;
        const int64_t p0 = crow[r], p1 = crow[r + 1];
        for (int64_t c4 = lir; c4 < D4; c4 += LPR) {
            const float4 acc = spmm_row_range<LPR>(col, val, X, ncols, D, c4, p0, p1, 1);
            spmm_store<LPR>(acc, r, D, c4, Y, Z, beta, ACC, acc_scale);
        }
    }
}

// long rows: one workgroup per CHUNK of SP_CHUNK non-zeros (a popular item can have tens of thousands of non-zeros: one
// workgroup per row would be the tail of the whole launch).  Lane group j takes non-zeros p0 + j, p0 + j + G, ...; the 16
// group partials are combined in group order -> one partial row per chunk; spmm_csr_long_combine then adds a row's chunk
// partials in chunk order and applies the epilogue.  Fixed order everywhere => bitwise reproducible.
#define SP_CHUNK 2048
template <int LPR>
__global__ __launch_bounds__(256) void spmm_csr_long(const int64_t* __restrict__ row_order, const int32_t* __restrict__ chunk_row,
                                                     const int64_t* __restrict__ chunk_ptr, int64_t nchunks,
                                                     const int64_t* __restrict__ crow, const int64_t* __restrict__ col,
                                                     const float* __restrict__ val, int64_t ncols, const float* __restrict__ X,
                                                     int64_t D, float* __restrict__ partial) {
    __shared__ float4 part[256];
    const int lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    constexpr int G = 256 / LPR;
    const int64_t D4 = D >> 2;
    for (int64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        const int li = chunk_row[ch];
        const int64_t r = row_order[li];
        const int64_t p0 = crow[r] + (ch - chunk_ptr[li]) * SP_CHUNK;
        const int64_t p1 = (p0 + SP_CHUNK < crow[r + 1]) ? p0 + SP_CHUNK : crow[r + 1];
        for (int64_t c4 = lir; c4 < D4; c4 += LPR) {
            part[threadIdx.x] = spmm_row_range<LPR>(col, val, X, ncols, D, c4, p0 + grp, p1, G);
            __syncthreads();
            if (grp == 0) {
                float4 acc = part[lir];
                for (int j = 1; j < G; ++j) {
                    const float4 q = part[j * LPR + lir];
                    acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
                }
                reinterpret_cast<float4*>(partial + ch * D)[c4] = acc;
            }
            __syncthreads();
        }
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void spmm_csr_long_combine(const int64_t* __restrict__ row_order, int64_t nlong,
                                                             const int64_t* __restrict__ chunk_ptr, const float* __restrict__ partial,
                                                             int64_t D, float* __restrict__ Y, const float* __restrict__ Z, float beta,
                                                             float* __restrict__ ACC, float acc_scale) {
    const int lir = threadIdx.x % LPR;
    const int64_t li = (int64_t)blockIdx.x * (256 / LPR) + threadIdx.x / LPR;
    if (li >= nlong) return;
    const int64_t r = row_order[li];
    const int64_t D4 = D >> 2;
    for (int64_t c4 = lir; c4 < D4; c4 += LPR) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int64_t ch = chunk_ptr[li]; ch < chunk_ptr[li + 1]; ++ch) {
            const float4 q = reinterpret_cast<const float4*>(partial + ch * D)[c4];
            acc.x += q.x; acc.y += q.y; acc.z += q.z; acc.w += q.w;
        }
        spmm_store<LPR>(acc, r, D, c4, Y, Z, beta, ACC, acc_scale);
    }
}

extern "C" int re_spmm_csr(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                           const int64_t* row_order, int64_t nlong, const int32_t* chunk_row, const int64_t* chunk_ptr,
                           int64_t nchunks, const float* X, int64_t D, float* Y, const float* Z, float beta, float* ACC,
                           float acc_scale, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (nrows == 0) return RE_OK;
    if (!crow || !col || !val || !X || !Y || nrows < 0 || ncols <= 0 || D <= 0 || nlong < 0 || nlong > nrows) return RE_EINVAL;
    if (nlong > 0 && (!row_order || !chunk_row || !chunk_ptr || nchunks < nlong || !ws)) return RE_EINVAL;
    if (nlong > 0 && ws_bytes < (size_t)nchunks * D * sizeof(float)) return RE_EWORKSPACE;
    if ((D & 3) || ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Z) |
                     reinterpret_cast<uintptr_t>(ACC) | reinterpret_cast<uintptr_t>(ws)) & 15u))
        return RE_EUNSUPPORTED;
    if (X == Y) return RE_EINVAL;  // not in place
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)ws;
#define SP_LAUNCH(LPRV)                                                                                                             \
    do {                                                                                                                            \
        if (nlong) hipLaunchKernelGGL(spmm_csr_long<LPRV>, dim3(re_grid(nchunks, 1, 65536)), dim3(256), 0, s, row_order, chunk_row, chunk_ptr, nchunks, crow, col, val, ncols, X, D, partial); \
        if (nrows > nlong) hipLaunchKernelGGL(spmm_csr_rows<LPRV>, dim3(re_grid(nrows - nlong, 256 / LPRV, 65536)), dim3(256), 0, s, row_order, nlong, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale); \
        if (nlong) hipLaunchKernelGGL(spmm_csr_long_combine<LPRV>, dim3((unsigned)re_cdiv(nlong, 256 / LPRV)), dim3(256), 0, s, row_order, nlong, chunk_ptr, partial, D, Y, Z, beta, ACC, acc_scale); \
    } while (0)
    if ((D >> 2) >= 32) SP_LAUNCH(32); else SP_LAUNCH(16);
#undef SP_LAUNCH
    return re_launch_status();
}

// rows' squared L2 norms: out[0] = scale * sum_i ||W[idx[i], :]||^2   (BaseCriterion.regularize(.., "l2") = sum/2,
// LightGCN/main.py:99-106).  Deterministic: block partials, then one wave adds them in order.
__global__ __launch_bounds__(256) void rows_sqnorm_k(const float* __restrict__ W, int64_t R, int64_t D, const int64_t* __restrict__ idx,
                                                     int64_t n, float* __restrict__ bsum) {
    __shared__ float s_sum[4];
    float acc = 0.f;
    const int64_t total = n * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t i = e / D, d = e - i * D;
        const int64_t r = idx[i];
        if (r >= 0 && r < R) { const float x = W[r * D + d]; acc = fmaf(x, x, acc); }
    }
    acc = re_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_sum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
}
__global__ __launch_bounds__(64) void rows_sqnorm_fin(const float* __restrict__ bsum, int nb, float scale, float* __restrict__ out, int accumulate) {
    float s = 0.f;
    for (int b = threadIdx.x; b < nb; b += 64) s += bsum[b];
    s = re_wave_sum(s);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * scale;
}

extern "C" size_t re_rows_sqnorm_workspace_bytes(void) { return 1024 * sizeof(float); }
extern "C" int re_rows_sqnorm(const float* W, int64_t R, int64_t D, const int64_t* idx, int64_t n, float scale, float* out,
                              int accumulate, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!W || !idx || !out || !ws || n < 0 || R <= 0 || D <= 0) return RE_EINVAL;
    if (ws_bytes < re_rows_sqnorm_workspace_bytes()) return RE_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)re_grid(n * D, 256 * 4, 1024);
    hipLaunchKernelGGL(rows_sqnorm_k, dim3(nb), dim3(256), 0, s, W, R, D, idx, n, (float*)ws);
    hipLaunchKernelGGL(rows_sqnorm_fin, dim3(1), dim3(64), 0, s, (const float*)ws, nb, scale, out, accumulate);
    return re_launch_status();
}
