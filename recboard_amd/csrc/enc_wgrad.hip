// Weight gradients of the SASRec encoder, after enc_bwd.hip: dW = dY^T X for the six linear maps of every block, as split-K
// products over the batch's compact rows (the contraction runs over ALL rows, ~3 600 on a Beauty-shaped batch): grid =
// (row splits) x (6 matrices) x (blocks), every workgroup accumulates one D x D partial over its share of the row tiles
// (operands are contiguous row ranges of the tape / gradient tape, staged through LDS), `enc_grad_reduce_k` adds the partials
// and the workgroups' vector-gradient slabs in a fixed order (deterministic, no float atomics).  The position-table gradient
// (sum over the batch of the embedding-backward rows at every position, SASRec/main.py:183 `+ Position(positions)`) rides in
// the same launch as one more grid plane.
//   gradient-tape order: 0 dO2 (x HR -> W2)  1 dH (x Y -> W1)  2 dX1 (x O -> Wo)  3 dQ (x A -> Wq)  4 dK (x X -> Wk)  5 dV (x X -> Wv)
#include "enc_common.h"

#define WG_NSPLIT 24
#define WG_CH 4   // row tiles per LDS stage

size_t enc_wgrad_part_floats(int64_t D, int64_t L) { return (size_t)L * EG_NMAT * WG_NSPLIT * D * D; }

template <int D>
__global__ __launch_bounds__(512) void enc_wgrad_k(const float* __restrict__ tape, EncTape T, const float* __restrict__ gtape, int64_t NR,
                                                   const void* __restrict__ planp, int B, int S, int L, float* __restrict__ part,
                                                   const int64_t* __restrict__ seq, const float* __restrict__ contrib, float* __restrict__ ppart,
                                                   const float* __restrict__ dPtab) {
    using C = EC<D>;
    constexpr int RTW = C::NS / C::WR;   // output row tiles per wave
    extern __shared__ __align__(16) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wave % C::NS, wr = wave / C::NS, g = lane >> 4, c = lane & 15;
    if ((int)blockIdx.z == L) {
        // ---- position-table gradient, partial sums: job (p, chunk of 64 sequences) -> ppart[p][chunk][D] = sum over the chunk's
        //      sequences with a real token at p of contrib[b][p]  (every load independent: one memory round trip per job)
        if (!dPtab) return;
        const int nch = (B + 63) / 64, njobs = S * nch;
        const int col = tid % D, rg = tid / D;
        // (the next job's loads are in flight while this one is reduced: a workgroup has ~3 jobs, each a memory round trip)
        constexpr int NQP = 64 / C::CG;
        int64_t sv[NQP], svn[NQP];
        float cv[NQP], cvn[NQP];
        const int jstep = gridDim.x * gridDim.y, j0 = blockIdx.y * gridDim.x + blockIdx.x;
#define WG_PJOB(J, SV, CV)                                                              \
        do {                                                                            \
            const int p_ = (J) / nch, ch_ = (J) % nch;                                  \
            _Pragma("unroll") for (int q = 0; q < NQP; ++q) {                           \
                int b = ch_ * 64 + rg + C::CG * q;                                      \
                const bool in_ = b < B;                                                 \
                b = in_ ? b : B - 1;   /* clamped, unconditional loads */               \
                const int64_t sx = seq[(int64_t)b * S + p_];                            \
                const float cx = contrib[((int64_t)b * S + p_) * D + col];              \
                SV[q] = in_ ? sx : 0;                                                   \
                CV[q] = cx;                                                             \
            }                                                                           \
        } while (0)
        if (j0 < njobs) WG_PJOB(j0, sv, cv);
        for (int j = j0; j < njobs; j += jstep) {
            if (j + jstep < njobs) WG_PJOB(j + jstep, svn, cvn);
            float s = 0.f;
#pragma unroll
            for (int q = 0; q < NQP; ++q) s += (sv[q] != 0) ? cv[q] : 0.f;   // (rows of pad positions are never written: select, not multiply)
            __syncthreads();
            lds[tid] = s;
            __syncthreads();
            if (tid < D) {
                float t = lds[tid];
#pragma unroll
                for (int i = 1; i < C::CG; ++i) t += lds[i * D + tid];
                ppart[(int64_t)j * D + tid] = t;
            }
#pragma unroll
            for (int q = 0; q < NQP; ++q) { sv[q] = svn[q]; cv[q] = cvn[q]; }
        }
#undef WG_PJOB
        return;
    }
    const int l = blockIdx.z, m = blockIdx.y, split = blockIdx.x;
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_tiles = PL.hdr[1];
    const int per = (n_tiles + WG_NSPLIT - 1) / WG_NSPLIT;
    const int t0 = split * per;
    const int t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    const int64_t xoff = (m == 0) ? T.off_HR : (m == 1) ? T.off_Y : (m == 2) ? T.off_O : (m == 3) ? T.off_A : T.off_X;
    const float* X = tape + (int64_t)l * T.per_block + xoff;
    const float* dY = gtape + ((int64_t)l * EG_NMAT + m) * NR * D;
    float* bufA = lds;
    float* bufB = lds + 16 * WG_CH * C::LS;
    f32x4 acc[RTW];
#pragma unroll
    for (int t = 0; t < RTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // stages of WG_CH row tiles through LDS; the next stage's rows are requested into registers before this stage's products
    // (every stage is a memory round trip: four of them in a row were most of the launch)
    constexpr int NQ = 16 * WG_CH * (D / 4) / C::NT;
    f32x4 ra[NQ], rb[NQ];   // (native vectors and a macro: HIP's float4 struct arrays / arrays captured by a lambda stay in scratch memory)
#define WG_FETCH(TC)                                                                                            \
    do {                                                                                                        \
        const int ntc_ = (t1 - (TC)) < WG_CH ? (t1 - (TC)) : WG_CH;                                             \
        const int nf_ = 16 * ntc_ * (D / 4);                                                                    \
        _Pragma("unroll") for (int q = 0; q < NQ; ++q) {                                                        \
            int f = q * C::NT + tid;                                                                            \
            f = f < nf_ ? f : nf_ - 1;   /* clamped, unconditional: a predicated load is waited for on the spot */ \
            ra[q] = reinterpret_cast<const f32x4*>(dY + (int64_t)(TC) * 16 * D)[f];                             \
            rb[q] = reinterpret_cast<const f32x4*>(X + (int64_t)(TC) * 16 * D)[f];                              \
        }                                                                                                       \
    } while (0)
    if (t0 < t1) WG_FETCH(t0);
    for (int tc = t0; tc < t1; tc += WG_CH) {
        const int ntc = (t1 - tc) < WG_CH ? (t1 - tc) : WG_CH;
        const int nf = 16 * ntc * (D / 4);
        enc_sync();
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = q * C::NT + tid;
            if (f < nf) {
                const int r = f / (D / 4), c4 = f % (D / 4);
                *reinterpret_cast<f32x4*>(bufA + r * C::LS + 4 * c4) = ra[q];
                *reinterpret_cast<f32x4*>(bufB + r * C::LS + 4 * c4) = rb[q];
            }
        }
        if (tc + WG_CH < t1) WG_FETCH(tc + WG_CH);
        enc_sync();
        for (int q = 0; q < ntc; ++q) {
            float bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = bufB[(16 * q + 4 * g + i) * C::LS + 16 * strip + c];
#pragma unroll
            for (int t = 0; t < RTW; ++t) {
                const int mt = t * C::WR + wr;
                float af[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) af[i] = bufA[(16 * q + 4 * g + i) * C::LS + 16 * mt + c];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[i], acc[t], 0, 0, 0);
            }
        }
    }
#undef WG_FETCH
    float* out = part + (((int64_t)l * EG_NMAT + m) * WG_NSPLIT + split) * D * D;
#pragma unroll
    for (int t = 0; t < RTW; ++t) {
        const int mt = t * C::WR + wr;
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(16 * mt + 4 * g + j) * D + 16 * strip + c] = acc[t][j];
    }
}

// optional: the dense Adam of every gradient element the reduction finishes (re_adam_fuse: arenas of one layout)
struct EncAdam {
    const float* gbase;
    float *p, *m, *v;
    const float* hyper;
    float b1, b2, omb1, omb2, eps, wd;
};
__device__ __forceinline__ void eg_put(const EncAdam& A, float* d, float g) {
    *d = g;
    if (A.p) {
        const float ss = A.hyper[0], ib = A.hyper[1];
        if (ib != 0.f) {   // ({0, 0}: the caller gated this step off)
            const int64_t i = d - A.gbase;
            float pp = A.p[i], mm = A.m[i], vv = A.v[i];
            re_adam1(pp, mm, vv, g, A.b1, A.b2, A.omb1, A.omb2, ss, ib, A.eps, A.wd);     // (adam_vec4_dev's arithmetic)
            A.p[i] = pp; A.m[i] = mm; A.v[i] = vv;
        }
    }
}

struct EncGradDst {
    float* p[SE_MAX_BLOCKS][14];  // per block: ABI order of the 12 block gradients, then g_last_w, g_last_b (last block only)
};

// blocks [0, nmat_blocks): 256 elements of the L * 6 * D * D weight gradients each (sum of the WG_NSPLIT partials);
// the rest: 64 columns of one (block, vector) each, summed over the slabs of the workgroups that had work (4 waves x fixed order).
__global__ __launch_bounds__(256) void enc_grad_reduce_k(const float* __restrict__ part, const float* __restrict__ slab, int nwg,
                                                         const void* __restrict__ planp, int B, int S, int D, int L, EncGradDst dst,
                                                         int nmat_blocks, int nvec_blocks, const float* __restrict__ ppart, float inv_scale,
                                                         float* __restrict__ dPtab, int by_tile, EncAdam AD) {
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nmat_blocks + nvec_blocks) {
        // position-table gradient: the chunk partials of enc_wgrad_k in chunk order, / scale
        const int e = ((int)blockIdx.x - nmat_blocks - nvec_blocks) * 256 + tid;
        if (e >= S * D) return;
        const int p = e / D, cc = e % D, nch = (B + 63) / 64;
        float s = 0.f;
        for (int ch = 0; ch < nch; ++ch) s += ppart[((int64_t)p * nch + ch) * D + cc];
        eg_put(AD, dPtab + e, s * inv_scale);
        return;
    }
    if ((int)blockIdx.x < nmat_blocks) {
        const int64_t e = (int64_t)blockIdx.x * 256 + tid;
        const int dd = D * D;
        if (e >= (int64_t)L * EG_NMAT * dd) return;
        const int lm = (int)(e / dd), off = (int)(e % dd);
        const float* p = part + (int64_t)lm * WG_NSPLIT * dd + off;
        float v[WG_NSPLIT];
#pragma unroll
        for (int i = 0; i < WG_NSPLIT; ++i) v[i] = p[(int64_t)i * dd];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WG_NSPLIT; ++i) s += v[i];
        const int l = lm / EG_NMAT, m = lm % EG_NMAT;
        float* const* P = dst.p[l];
        float* d = (m == 0) ? P[10] : (m == 1) ? P[8] : (m == 2) ? P[4] : P[2] + (m - 3) * dd;
        eg_put(AD, d + off, s);
        return;
    }
    __shared__ float red[4][64];
    const int job = blockIdx.x - nmat_blocks;           // (l, v, 64-column group)
    const int cgs = D / 64;
    const int cg = job % cgs, v = (job / cgs) % EG_NVEC, l = job / (cgs * EG_NVEC);
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_items = PL.hdr[0];
    // slab rows: one per workgroup that had work, or -- D = 64 steps that ran one tile per workgroup (enc_tile.hip) -- one per tile
    const int nact = (by_tile && PL.hdr[7] == 1) ? PL.hdr[1] : (n_items < nwg ? n_items : nwg);
    const int lane = tid & 63, wave = tid >> 6;
    const float* sl = slab + ((int64_t)l * EG_NVEC + v) * D + cg * 64 + lane;
    const int64_t stride = (int64_t)L * EG_NVEC * D;
    float s = 0.f;
    for (int w0 = wave; w0 < nact; w0 += 32) {
        float x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int w = w0 + 4 * q;
            x[q] = (w < nact) ? sl[(int64_t)w * stride] : 0.f;
        }
        s += ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave != 0) return;
    s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    if (v >= 10 && l != L - 1) return;
    float* const* P = dst.p[l];
    float* d;
    switch (v) {
        case 0: case 1: case 2: d = P[3] + v * D; break;
        case 3: d = P[5]; break;
        case 4: d = P[9]; break;
        case 5: d = P[11]; break;
        case 6: d = P[0]; break;
        case 7: d = P[1]; break;
        case 8: d = P[6]; break;
        case 9: d = P[7]; break;
        case 10: d = P[12]; break;
        default: d = P[13]; break;
    }
    eg_put(AD, d + cg * 64 + lane, s);
}

size_t enc_wgrad_ppart_floats(int64_t B, int64_t D) { return (size_t)64 * ((B + 63) / 64) * D; }

int enc_wgrad_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* tape, const float* gtape, const void* plan, const float* slab,
                     int nwg, float* part, float* ppart, const int64_t* seq, const float* contrib, float emb_scale, float* dPtab,
                     float* const* block_grads, float* g_last_w, float* g_last_b, hipStream_t s, int by_tile, const re_adam_fuse* adam) {
    if (D != 64 && D != 128) return RE_EUNSUPPORTED;
    const EncTape T = enc_tape_layout(B, S, D, L);
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    const dim3 grid(WG_NSPLIT, EG_NMAT, (unsigned)(L + (dPtab ? 1 : 0)));
    if (D == 128) {
        using C = EC<128>;
        const size_t ldsb = (size_t)2 * 16 * WG_CH * C::LS * sizeof(float);
        if (hipFuncSetAttribute((const void*)enc_wgrad_k<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(enc_wgrad_k<128>, grid, dim3(C::NT), ldsb, s, (const float*)tape, T, gtape, NR, plan, (int)B, (int)S, (int)L, part, seq, contrib,
                           ppart, (const float*)dPtab);
    } else {
        using C = EC<64>;
        const size_t ldsb = (size_t)2 * 16 * WG_CH * C::LS * sizeof(float);
        hipLaunchKernelGGL(enc_wgrad_k<64>, grid, dim3(C::NT), ldsb, s, (const float*)tape, T, gtape, NR, plan, (int)B, (int)S, (int)L, part, seq, contrib,
                           ppart, (const float*)dPtab);
    }
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    EncGradDst dst;
    for (int64_t l = 0; l < SE_MAX_BLOCKS; ++l)
        for (int i = 0; i < 14; ++i) dst.p[l][i] = (l < L && i < 12) ? block_grads[12 * l + i] : (i == 12 ? g_last_w : g_last_b);
    const int nmat_blocks = (int)((L * EG_NMAT * D * D + 255) / 256);
    const int nvec_blocks = (int)(L * EG_NVEC * (D / 64));
    const int npos_blocks = dPtab ? (int)((S * D + 255) / 256) : 0;
    EncAdam AD{};
    if (adam) {
        if (!adam->grad_base || !adam->param || !adam->m || !adam->v || !adam->hyper) return RE_EINVAL;
        AD = EncAdam{adam->grad_base, adam->param, adam->m, adam->v, adam->hyper, (float)adam->beta1, (float)adam->beta2, (float)(1.0 - adam->beta1),
                     (float)(1.0 - adam->beta2), (float)adam->eps, (float)adam->weight_decay};
    }
    hipLaunchKernelGGL(enc_grad_reduce_k, dim3(nmat_blocks + nvec_blocks + npos_blocks), dim3(256), 0, s, (const float*)part, slab, nwg, plan,
                       (int)B, (int)S, (int)D, (int)L, dst, nmat_blocks, nvec_blocks, (const float*)ppart, emb_scale != 0.f ? 1.0f / emb_scale : 0.f,
                       dPtab, by_tile, AD);
    return re_launch_status();
}
