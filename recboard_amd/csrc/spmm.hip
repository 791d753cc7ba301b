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
// NT: the CSR stream (col, val: 12 B per non-zero, each read ONCE per launch) and the output rows are moved with the non-temporal hint, so
// that what an XCD's 4 MiB L2 keeps are the gathered X rows -- the only bytes of the launch that are ever read twice.
template <bool NT, typename T>
__device__ __forceinline__ T sp_ld(const T* p) { if constexpr (NT) return __builtin_nontemporal_load(p); else return *p; }
template <bool NT>
__device__ __forceinline__ void sp_st4(float4* p, const float4& v) {
    if constexpr (NT) {
        float* q = reinterpret_cast<float*>(p);
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<f4v*>(q));
    } else *p = v;
}
template <bool NT>
__device__ __forceinline__ float4 sp_ld4(const float4* p) {
    if constexpr (NT) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v w = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
        return make_float4(w.x, w.y, w.z, w.w);
    } else return *p;
}

// MASK: `mask` holds one bit per row of X -- 0 = the row is all zeros and is not fetched (the FIRST product of LightGCN's backward pass: its input
// is the scatter of 3 B gradient rows into a 122 915-row array -- 95 % zero rows, and the launch is bound by the rows it gathers).  Adding an
// exact zero changes no sum: the result is the unmasked one bit for bit.  The mask words of a group are requested when its column ids
// arrive (16 KB for the Yelp2018 shape: L1 / L2 hits), in front of the next group's (col, val).
template <int LPR, bool NT = false, bool MASK = false>
__device__ __forceinline__ float4 spmm_row_range(const int64_t* __restrict__ col, const float* __restrict__ val,
                                                  const float* __restrict__ X, int64_t ncols, int64_t D, int64_t c4,
                                                  int64_t p0, int64_t p1, int64_t step, const uint32_t* __restrict__ mask = nullptr) {
    constexpr int U = SP_ILP;
    auto in_range = [&](int64_t c) { return c >= 0 && c < ncols; };
    auto mword = [&](int64_t c) { return MASK ? mask[(in_range(c) ? c : 0) >> 5] : 0xFFFFFFFFu; };      // (clamped, unconditional)
    auto live = [&](int64_t c, uint32_t w) { return in_range(c) && (!MASK || ((w >> (c & 31)) & 1u)); };
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int64_t p = p0;
    if (p + (U - 1) * step < p1) {
        int64_t cc[U];
        float vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { cc[u] = sp_ld<NT>(col + p + u * step); vv[u] = sp_ld<NT>(val + p + u * step); }
        for (;;) {
            const int64_t pn = p + U * step;
            const bool more = pn + (U - 1) * step < p1;
            uint32_t mw[U];
#pragma unroll
            for (int u = 0; u < U; ++u) mw[u] = mword(cc[u]);
            int64_t nc[U];
            float nv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {   // (clamped, unconditional: in flight beside the X rows below.  Padding the last group with
                const int64_t q = more ? pn + u * step : p;   //  weight-0 slots instead of the tail loops was measured slower: 136 / 156 vs 126 us)
                nc[u] = sp_ld<NT>(col + q); nv[u] = sp_ld<NT>(val + q);
            }
            float4 xr[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (live(cc[u], mw[u])) xr[u] = reinterpret_cast<const float4*>(X + cc[u] * D)[c4];
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
        for (int u = 0; u < 4; ++u) { cc[u] = sp_ld<NT>(col + p + u * step); vv[u] = sp_ld<NT>(val + p + u * step); }
        uint32_t mw[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) mw[u] = mword(cc[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (live(cc[u], mw[u])) xr[u] = reinterpret_cast<const float4*>(X + cc[u] * D)[c4];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) f4_axpy(acc, vv[u], xr[u]);
    }
    for (; p < p1; p += step) {
        const int64_t cc = sp_ld<NT>(col + p);
        if (live(cc, mword(cc))) f4_axpy(acc, sp_ld<NT>(val + p), reinterpret_cast<const float4*>(X + cc * D)[c4]);
    }
    return acc;
}

// A whole row by ONE lane group, its (col, val) fetched by the group instead of by every lane: lane l of the group loads non-zero p + l
// (one coalesced 8-byte and one 4-byte load per lane for LPR non-zeros -- the row_range form above issues a (col, val) pair per non-zero in
// EVERY lane, all lanes of a group the same address: 2 of the 3 memory instructions a non-zero costs, and the launch is bound by the CUs'
// address units as much as by the rows they fetch: masking 95 % of the gathers away saved 15 %), then hands them round with cross-lane moves.
// The next block's (col, val) are in flight while this block's X rows are gathered.  Sums in non-zero order: the same bits as spmm_row_range.
template <int LPR, bool NT = false, bool MASK = false>
__device__ __forceinline__ float4 spmm_row_contig(const int64_t* __restrict__ col, const float* __restrict__ val,
                                                   const float* __restrict__ X, int64_t ncols, int64_t D, int64_t c4,
                                                   int64_t p0, int64_t p1, const uint32_t* __restrict__ mask = nullptr) {
    const int lir = threadIdx.x % LPR;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p0 >= p1) return acc;
    auto fetch = [&](int64_t p, int& c, float& v, uint32_t& w) {     // (clamped, unconditional; a slot past the row: column -1 = no row)
        const int64_t q = p + lir < p1 ? p + lir : p1 - 1;
        const int64_t cq = sp_ld<NT>(col + q);
        const float vq = sp_ld<NT>(val + q);
        const bool ok = p + lir < p1 && cq >= 0 && cq < ncols;
        c = ok ? (int)cq : -1;
        v = ok ? vq : 0.f;
        w = 0xFFFFFFFFu;
        if (MASK) w = mask[(ok ? cq : 0) >> 5];                   // (behind the column id: an L1 / L2 hit)
    };
    int cm, cn = -1;
    float vm, vn = 0.f;
    uint32_t wm, wn = 0u;
    fetch(p0, cm, vm, wm);
    for (int64_t p = p0; p < p1; p += LPR) {
        fetch(p + LPR, cn, vn, wn);                                 // (always: past the row every slot is a clamped "no row" -- no branch around the loads)
#pragma unroll
        for (int h = 0; h < LPR; h += 8) {
            if (p + h >= p1) break;                                  // (uniform)
            int cc[8];
            float vv[8];
            float4 xr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                cc[u] = __shfl(cm, h + u, LPR);
                vv[u] = __shfl(vm, h + u, LPR);
                const uint32_t w = MASK ? (uint32_t)__shfl((int)wm, h + u, LPR) : 0xFFFFFFFFu;
                xr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cc[u] >= 0 && (!MASK || ((w >> (cc[u] & 31)) & 1u))) xr[u] = reinterpret_cast<const float4*>(X + (int64_t)cc[u] * D)[c4];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) f4_axpy(acc, vv[u], xr[u]);
        }
        cm = cn; vm = vn; wm = wn;
    }
    return acc;
}

template <int LPR, bool NT = false>
__device__ __forceinline__ void spmm_store(float4 acc, int64_t r, int64_t D, int64_t c4, float* __restrict__ Y,
                                           const float* __restrict__ Z, float beta, float* __restrict__ ACC, float acc_scale,
                                           const float* __restrict__ ainit = nullptr, const uint32_t* __restrict__ zmask = nullptr) {
    // (zmask: a bit per row of Z, 0 = that row is all zeros and is not read -- LightGCN's backward pass adds the SAME 95 %-empty scatter to
    //  all three products)
    if (Z && (!zmask || ((zmask[r >> 5] >> (r & 31)) & 1u))) f4_axpy(acc, beta, sp_ld4<NT>(reinterpret_cast<const float4*>(Z + r * D) + c4));
    sp_st4<NT>(reinterpret_cast<float4*>(Y + r * D) + c4, acc);
    if (ACC) {
        // (ainit: the running sum STARTS here -- acc_scale x the row of `ainit` (the first propagation's own input: LightGCN's layer-0 term)
        //  instead of what ACC holds: the launch that filled ACC with acc_scale * X0 beforehand is not needed)
        float4 a = sp_ld4<NT>(reinterpret_cast<const float4*>((ainit ? ainit : ACC) + r * D) + c4);
        if (ainit) { a.x *= acc_scale; a.y *= acc_scale; a.z *= acc_scale; a.w *= acc_scale; }
        f4_axpy(a, acc_scale, acc);
        sp_st4<NT>(reinterpret_cast<float4*>(ACC + r * D) + c4, a);
    }
}

// split > first (re_spmm_csr_split; the grid is then a multiple of 8): row_order[first, split) and row_order[split, nrows) are two ROW
// CLASSES that gather from different parts of X -- a bipartite adjacency's user rows gather item rows and the other way round -- and the
// blocks that share an XCD (blockIdx % 8 labels them: cdna_hip_programming.md T1) all walk ONE class: labels [0, k0) the first, [k0, 8) the
// second.  Each L2 then has to hold the hot rows of one part of X instead of both.
// (blk, nblk: this workgroup's index among the row-walking workgroups of the launch -- the fused launch below puts the long rows' chunk
//  workgroups in front of them)
template <int LPR, bool NT, bool MASK = false>
__device__ __forceinline__ void spmm_rows_walk(int64_t blk, int64_t nblk, const int64_t* __restrict__ row_order, int64_t first, int64_t split, int k0,
                                               const int64_t* __restrict__ crow, const int64_t* __restrict__ col,
                                               const float* __restrict__ val, int64_t nrows, int64_t ncols,
                                               const float* __restrict__ X, int64_t D, float* __restrict__ Y,
                                               const float* __restrict__ Z, float beta, float* __restrict__ ACC,
                                               float acc_scale, const float* __restrict__ ainit = nullptr,
                                               const uint32_t* __restrict__ mask = nullptr, const int64_t* __restrict__ row_ptrs = nullptr,
                                               const uint32_t* __restrict__ zmask = nullptr) {
    const int lir = threadIdx.x % LPR;
    const int64_t gpb = 256 / LPR;
    const int64_t D4 = D >> 2;
    int64_t lo = first, hi = nrows, bi = blk, nb = nblk;
    if (split > first) {
        const int lab = (int)(blk & 7), cls = lab >= k0;
        const int k = cls ? 8 - k0 : k0;
        bi = (blk >> 3) * k + (cls ? lab - k0 : lab);
        nb = (nblk >> 3) * k;
        lo = cls ? split : first;
        hi = cls ? nrows : split;
    }
    // rows are visited in descending-degree order (row_order): the 4 (or 2) rows a wave works on have similar lengths,
    // and the longest rows start first
    for (int64_t i = lo + bi * gpb + threadIdx.x / LPR; i < hi; i += nb * gpb) {
        // (row_ptrs: crow[row], crow[row + 1] in WALKING order -- read beside row_order[i] instead of behind it: a short row's chain of memory
        //  round trips is row id -> row pointers -> (col, val) -> X rows -> store, and a launch is ~7 rounds of such chains)
        const int64_t r = row_order ? row_order[i] : i;
        int64_t p0, p1;
        if (row_ptrs) { p0 = row_ptrs[2 * i]; p1 = row_ptrs[2 * i + 1]; }     // (uniform)
        else { p0 = crow[r]; p1 = crow[r + 1]; }
        for (int64_t c4 = lir; c4 < D4; c4 += LPR) {
            const float4 acc = spmm_row_contig<LPR, NT, MASK>(col, val, X, ncols, D, c4, p0, p1, mask);
            spmm_store<LPR, NT>(acc, r, D, c4, Y, Z, beta, ACC, acc_scale, ainit, zmask);
        }
    }
}
template <int LPR, bool NT, bool MASK = false>
__global__ __launch_bounds__(256) void spmm_csr_rows(const int64_t* __restrict__ row_order, int64_t first, int64_t split, int k0,
                                                     const int64_t* __restrict__ crow, const int64_t* __restrict__ col,
                                                     const float* __restrict__ val, int64_t nrows, int64_t ncols,
                                                     const float* __restrict__ X, int64_t D, float* __restrict__ Y,
                                                     const float* __restrict__ Z, float beta, float* __restrict__ ACC,
                                                     float acc_scale, int acc_init, const uint32_t* __restrict__ mask,
                                                     const uint32_t* __restrict__ zmask) {
    spmm_rows_walk<LPR, NT, MASK>(blockIdx.x, gridDim.x, row_order, first, split, k0, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale,
                                  acc_init ? X : nullptr, mask, nullptr, zmask);
}

// long rows: one workgroup per CHUNK of SP_CHUNK non-zeros (a popular item can have tens of thousands of non-zeros: one
// workgroup per row would be the tail of the whole launch).  Lane group j takes non-zeros p0 + j, p0 + j + G, ...; the 16
// group partials are combined in group order -> one partial row per chunk; spmm_csr_long_combine then adds a row's chunk
// partials in chunk order and applies the epilogue.  Fixed order everywhere => bitwise reproducible.
#define SP_CHUNK 2048
template <int LPR, bool MASK = false>
__device__ __forceinline__ void spmm_long_chunks(float4* part, int64_t first_chunk, int64_t chunk_stride,
                                                 const int64_t* __restrict__ row_order, const int32_t* __restrict__ chunk_row,
                                                 const int64_t* __restrict__ chunk_ptr, int64_t nchunks,
                                                 const int64_t* __restrict__ crow, const int64_t* __restrict__ col,
                                                 const float* __restrict__ val, int64_t ncols, const float* __restrict__ X,
                                                 int64_t D, float* __restrict__ partial, const uint32_t* __restrict__ mask = nullptr) {
    const int lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    constexpr int G = 256 / LPR;
    const int64_t D4 = D >> 2;
    for (int64_t ch = first_chunk; ch < nchunks; ch += chunk_stride) {
        const int li = chunk_row[ch];
        const int64_t r = row_order[li];
        const int64_t p0 = crow[r] + (ch - chunk_ptr[li]) * SP_CHUNK;
        const int64_t p1 = (p0 + SP_CHUNK < crow[r + 1]) ? p0 + SP_CHUNK : crow[r + 1];
        for (int64_t c4 = lir; c4 < D4; c4 += LPR) {
            part[threadIdx.x] = spmm_row_range<LPR, false, MASK>(col, val, X, ncols, D, c4, p0 + grp, p1, G, mask);
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

template <int LPR, bool MASK = false>
__global__ __launch_bounds__(256) void spmm_csr_long(const int64_t* __restrict__ row_order, const int32_t* __restrict__ chunk_row,
                                                     const int64_t* __restrict__ chunk_ptr, int64_t nchunks,
                                                     const int64_t* __restrict__ crow, const int64_t* __restrict__ col,
                                                     const float* __restrict__ val, int64_t ncols, const float* __restrict__ X,
                                                     int64_t D, float* __restrict__ partial, const uint32_t* __restrict__ mask) {
    __shared__ float4 part[256];
    spmm_long_chunks<LPR, MASK>(part, blockIdx.x, gridDim.x, row_order, chunk_row, chunk_ptr, nchunks, crow, col, val, ncols, X, D, partial, mask);
}

// ONE launch for the long rows' chunks AND the short rows (round 6): the first nch8 workgroups (nchunks rounded up to a multiple of 8, so that
// blockIdx % 8 -- the XCD label -- of the row walkers is unchanged) each take a chunk, the rest walk the short rows.  As launches of their own the
// chunk kernel's 459 workgroups (Yelp2018 shapes: 340 long rows) ran 22 us by themselves in front of the 100 us row walk, six times a step.
template <int LPR, bool NT, bool MASK = false>
__global__ __launch_bounds__(256) void spmm_csr_fused(const int32_t* __restrict__ chunk_row, const int64_t* __restrict__ chunk_ptr, int64_t nchunks,
                                                      int64_t nch8, float* __restrict__ partial,
                                                      const int64_t* __restrict__ row_order, int64_t first, int64_t split, int k0,
                                                      const int64_t* __restrict__ crow, const int64_t* __restrict__ col,
                                                      const float* __restrict__ val, int64_t nrows, int64_t ncols,
                                                      const float* __restrict__ X, int64_t D, float* __restrict__ Y,
                                                      const float* __restrict__ Z, float beta, float* __restrict__ ACC,
                                                      float acc_scale, int acc_init, int* __restrict__ arrived,
                                                      const uint32_t* __restrict__ mask, const int64_t* __restrict__ row_ptrs,
                                                      const uint32_t* __restrict__ zmask) {
    __shared__ float4 part[256];
    __shared__ int s_last;
    if ((int64_t)blockIdx.x < nch8) {
        spmm_long_chunks<LPR, MASK>(part, blockIdx.x, nch8, row_order, chunk_row, chunk_ptr, nchunks, crow, col, val, ncols, X, D, partial, mask);
        // `arrived` (a zeroed word per long row): the workgroup that brings a row's LAST chunk adds the row's chunk partials -- in chunk order,
        // whoever it is: the same sum -- and applies the epilogue; spmm_csr_long_combine, a launch of 6 us behind every propagation (six a
        // LightGCN step), is then not needed.  The others' partials come through their L2s' write-backs (release) and coherent loads here.
        if (!arrived || (int64_t)blockIdx.x >= nchunks) return;        // (nch8 <= 4096 <= the grid: one chunk per workgroup)
        const int li = chunk_row[blockIdx.x];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int n_of_row = (int)(chunk_ptr[li + 1] - chunk_ptr[li]);
            const int old = __hip_atomic_fetch_add(arrived + li, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == n_of_row - 1;
            if (s_last) __hip_atomic_store(arrived + li, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (!s_last) return;
        const int64_t r = row_order[li];
        for (int64_t c = threadIdx.x; c < D; c += 256) {
            float acc = 0.f;
            for (int64_t ch = chunk_ptr[li]; ch < chunk_ptr[li + 1]; ++ch)
                acc += __hip_atomic_load(partial + ch * D + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (spmm_store's arithmetic, a column at a time)
            if (Z && (!zmask || ((zmask[r >> 5] >> (r & 31)) & 1u))) acc = fmaf(beta, Z[r * D + c], acc);
            Y[r * D + c] = acc;
            if (ACC) {
                float a = acc_init ? X[r * D + c] * acc_scale : ACC[r * D + c];
                ACC[r * D + c] = fmaf(acc_scale, acc, a);
            }
        }
        return;
    }
    spmm_rows_walk<LPR, NT, MASK>((int64_t)blockIdx.x - nch8, (int64_t)gridDim.x - nch8, row_order, first, split, k0, crow, col, val, nrows, ncols, X, D,
                                  Y, Z, beta, ACC, acc_scale, acc_init ? X : nullptr, mask, row_ptrs, zmask);
}

template <int LPR>
__global__ __launch_bounds__(256) void spmm_csr_long_combine(const int64_t* __restrict__ row_order, int64_t nlong,
                                                             const int64_t* __restrict__ chunk_ptr, const float* __restrict__ partial,
                                                             int64_t D, float* __restrict__ Y, const float* __restrict__ Z, float beta,
                                                             float* __restrict__ ACC, float acc_scale, const float* __restrict__ ainit,
                                                             const uint32_t* __restrict__ zmask) {
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
        spmm_store<LPR>(acc, r, D, c4, Y, Z, beta, ACC, acc_scale, ainit, zmask);
    }
}

static int spmm_launch(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                       const int64_t* row_order, int64_t nlong, int64_t split, int k0, int flags, const int32_t* chunk_row, const int64_t* chunk_ptr,
                       int64_t nchunks, const float* X, int64_t D, float* Y, const float* Z, float beta, float* ACC,
                       float acc_scale, void* ws, size_t ws_bytes, re_stream_t stream, const uint32_t* mask = nullptr,
                       const int64_t* row_ptrs = nullptr) {
    re_clear_error();
    if (nrows == 0) return RE_OK;
    if (!crow || !col || !val || !X || !Y || nrows < 0 || ncols <= 0 || D <= 0 || nlong < 0 || nlong > nrows) return RE_EINVAL;
    if (nlong > 0 && (!row_order || !chunk_row || !chunk_ptr || nchunks < nlong || !ws)) return RE_EINVAL;
    // flags: 1 non-temporal streams; 2 the long rows' chunk partials are combined inside the launch -- `ws` then carries nlong int32 counters
    // behind the partials (zero before the first use; every call leaves them zero); 4 ACC starts at acc_scale * X[row] (square adjacency)
    const bool in_launch = (flags & 2) && nlong > 0;
    const int acc_init = (flags & 4) && ACC ? 1 : 0;
    // 8: the mask also names Z's non-zero rows (rows with a zero bit are not read); 16: the mask is for Z only (X is fetched in full)
    const uint32_t* zmask = (flags & 8) && mask && Z ? mask : (const uint32_t*)nullptr;
    if (flags & 16) mask = nullptr;
    if (acc_init && nrows != ncols) return RE_EINVAL;
    const size_t part_bytes = re_align((size_t)nchunks * D * sizeof(float), 16);
    if (nlong > 0 && ws_bytes < (in_launch ? part_bytes + (size_t)nlong * sizeof(int) : (size_t)nchunks * D * sizeof(float))) return RE_EWORKSPACE;
    if ((D & 3) || ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Y) | reinterpret_cast<uintptr_t>(Z) |
                     reinterpret_cast<uintptr_t>(ACC) | reinterpret_cast<uintptr_t>(ws)) & 15u))
        return RE_EUNSUPPORTED;
    if (X == Y) return RE_EINVAL;  // not in place
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)ws;
    int* arrived = in_launch ? (int*)((char*)ws + part_bytes) : (int*)nullptr;
    const float* ainit = acc_init ? X : (const float*)nullptr;
#define SP_LAUNCH(LPRV)                                                                                                             \
    do {                                                                                                                            \
        const int64_t nch8 = (nchunks + 7) & ~(int64_t)7;                                                                           \
        const bool fuse = nlong > 0 && nrows > nlong && nch8 <= 4096;                                                               \
        if (nlong && !fuse) {                                                                                                       \
            if (mask) hipLaunchKernelGGL((spmm_csr_long<LPRV, true>), dim3(re_grid(nchunks, 1, 65536)), dim3(256), 0, s, row_order, chunk_row, chunk_ptr, nchunks, crow, col, val, ncols, X, D, partial, mask); \
            else hipLaunchKernelGGL((spmm_csr_long<LPRV, false>), dim3(re_grid(nchunks, 1, 65536)), dim3(256), 0, s, row_order, chunk_row, chunk_ptr, nchunks, crow, col, val, ncols, X, D, partial, mask); \
        }                                                                                                                           \
        if (fuse) {                                                                                                                 \
            unsigned g = (unsigned)re_grid(nrows - nlong, 256 / LPRV, 65536 - 4096);                                                \
            g = (g + 7u) & ~7u;                                                                                                     \
            if (mask) hipLaunchKernelGGL((spmm_csr_fused<LPRV, false, true>), dim3(g + (unsigned)nch8), dim3(256), 0, s, chunk_row, chunk_ptr, nchunks, nch8, partial, row_order, nlong, split, k0, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale, acc_init, arrived, mask, row_ptrs, zmask); \
            else if (flags & 1) hipLaunchKernelGGL((spmm_csr_fused<LPRV, true>), dim3(g + (unsigned)nch8), dim3(256), 0, s, chunk_row, chunk_ptr, nchunks, nch8, partial, row_order, nlong, split, k0, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale, acc_init, arrived, mask, row_ptrs, zmask); \
            else hipLaunchKernelGGL((spmm_csr_fused<LPRV, false>), dim3(g + (unsigned)nch8), dim3(256), 0, s, chunk_row, chunk_ptr, nchunks, nch8, partial, row_order, nlong, split, k0, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale, acc_init, arrived, mask, row_ptrs, zmask); \
        } else if (nrows > nlong) {                                                                                                 \
            unsigned g = (unsigned)re_grid(nrows - nlong, 256 / LPRV, 65536);                                                       \
            if (split > nlong) g = (g + 7u) & ~7u;                                                                                  \
            if (mask) hipLaunchKernelGGL((spmm_csr_rows<LPRV, false, true>), dim3(g), dim3(256), 0, s, row_order, nlong, split, k0, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale, acc_init, mask, zmask); \
            else if (flags & 1) hipLaunchKernelGGL((spmm_csr_rows<LPRV, true>), dim3(g), dim3(256), 0, s, row_order, nlong, split, k0, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale, acc_init, mask, zmask); \
            else hipLaunchKernelGGL((spmm_csr_rows<LPRV, false>), dim3(g), dim3(256), 0, s, row_order, nlong, split, k0, crow, col, val, nrows, ncols, X, D, Y, Z, beta, ACC, acc_scale, acc_init, mask, zmask); \
        }                                                                                                                           \
        if (nlong && !(fuse && arrived)) hipLaunchKernelGGL(spmm_csr_long_combine<LPRV>, dim3((unsigned)re_cdiv(nlong, 256 / LPRV)), dim3(256), 0, s, row_order, nlong, chunk_ptr, partial, D, Y, Z, beta, ACC, acc_scale, ainit, zmask); \
    } while (0)
    if ((D >> 2) >= 32) SP_LAUNCH(32); else SP_LAUNCH(16);
#undef SP_LAUNCH
    return re_launch_status();
}

extern "C" int re_spmm_csr(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                           const int64_t* row_order, int64_t nlong, const int32_t* chunk_row, const int64_t* chunk_ptr,
                           int64_t nchunks, const float* X, int64_t D, float* Y, const float* Z, float beta, float* ACC,
                           float acc_scale, void* ws, size_t ws_bytes, re_stream_t stream) {
    return spmm_launch(crow, col, val, nrows, ncols, row_order, nlong, 0, 0, 0, chunk_row, chunk_ptr, nchunks, X, D, Y, Z, beta, ACC, acc_scale, ws,
                       ws_bytes, stream);
}

// The same product with the plan's two row classes kept apart by XCD (spmm_csr_rows) and / or the once-read streams moved non-temporally:
// row_order = [nlong long rows | class 0 | class 1], split = index of class 1's first row (<= nlong: one class), xcd_share = how many of the
// 8 XCD labels walk class 0 (1 .. 7), flags & 1 = non-temporal streams.  Results are bit-identical to re_spmm_csr's: a row's sum does not
// depend on which workgroup computes it.
extern "C" int re_spmm_csr_split(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                                 const int64_t* row_order, int64_t nlong, int64_t split, int32_t xcd_share, int32_t flags,
                                 const int32_t* chunk_row, const int64_t* chunk_ptr, int64_t nchunks, const float* X, int64_t D, float* Y,
                                 const float* Z, float beta, float* ACC, float acc_scale, void* ws, size_t ws_bytes, re_stream_t stream) {
    if (split < 0 || split > nrows || (split > nlong && (!row_order || xcd_share < 1 || xcd_share > 7))) { re_clear_error(); return RE_EINVAL; }
    return spmm_launch(crow, col, val, nrows, ncols, row_order, nlong, split > nlong && split < nrows ? split : 0, xcd_share, flags, chunk_row,
                       chunk_ptr, nchunks, X, D, Y, Z, beta, ACC, acc_scale, ws, ws_bytes, stream);
}

// re_spmm_csr_split with a bit per row of X: 0 = the row is all zeros (it is then not fetched; the result is the unmasked one bit for bit),
// and / or the rows' (crow[r], crow[r + 1]) pairs in row_order's order (row_ptrs [nrows][2], optional: used by the fused launch only).  The masks
// are honoured by every form of the launch: a caller may leave the masked-out rows of X / Z unwritten.
extern "C" int re_spmm_csr_masked(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                                  const int64_t* row_order, int64_t nlong, int64_t split, int32_t xcd_share, int32_t flags,
                                  const int32_t* chunk_row, const int64_t* chunk_ptr, int64_t nchunks, const float* X, int64_t D, float* Y,
                                  const float* Z, float beta, float* ACC, float acc_scale, const uint32_t* src_mask, const int64_t* row_ptrs,
                                  void* ws, size_t ws_bytes, re_stream_t stream) {
    if (split < 0 || split > nrows || (split > nlong && (!row_order || xcd_share < 1 || xcd_share > 7))) { re_clear_error(); return RE_EINVAL; }
    return spmm_launch(crow, col, val, nrows, ncols, row_order, nlong, split > nlong && split < nrows ? split : 0, xcd_share, flags, chunk_row,
                       chunk_ptr, nchunks, X, D, Y, Z, beta, ACC, acc_scale, ws, ws_bytes, stream, src_mask, row_ptrs);
}

// mask[i >> 5] bit (i & 31) = 1 for every i in rows[0 .. n) inside [0, nbits), 0 elsewhere: which rows of a scatter's dense output can be
// non-zero.  One workgroup (the mask of 122 915 rows is 15 KB): clear, barrier, OR.
__global__ __launch_bounds__(1024) void row_mask_k(const int64_t* __restrict__ rows, int64_t n, int64_t nbits, uint32_t* __restrict__ mask) {
    const int64_t nw = (nbits + 31) >> 5;
    for (int64_t i = threadIdx.x; i < nw; i += 1024) mask[i] = 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const int64_t r = rows[i];
        if (r >= 0 && r < nbits) atomicOr(mask + (r >> 5), 1u << (r & 31));
    }
}
extern "C" int re_row_mask(const int64_t* rows, int64_t n, int64_t nbits, uint32_t* mask, re_stream_t stream) {
    re_clear_error();
    if (!mask || nbits <= 0 || n < 0 || (n > 0 && !rows)) return RE_EINVAL;
    hipLaunchKernelGGL(row_mask_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, rows, n, nbits, mask);
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
