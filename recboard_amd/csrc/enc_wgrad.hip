// Weight gradients of the SASRec encoder, after enc_bwd.hip: dW = dY^T X for the six linear maps of every block, as split-K
// products over the batch's compact rows (the contraction runs over ALL rows, ~3 600 on a Beauty-shaped batch): grid =
// (row splits) x (6 matrices) x (blocks), every workgroup accumulates one D x D partial over its share of the row tiles
// (operands are contiguous row ranges of the tape / gradient tape, staged through LDS), `enc_grad_reduce_k` adds the partials
// and the workgroups' vector-gradient slabs in a fixed order (deterministic, no float atomics).  The position-table gradient
// (sum over the batch of the embedding-backward rows at every position, SASRec/main.py:183 `+ Position(positions)`) rides in
// the same launch as one more grid plane.
//   gradient-tape order: 0 dO2 (x HR -> W2)  1 dH (x Y -> W1)  2 dX1 (x O -> Wo)  3 dQ (x A -> Wq)  4 dK (x X -> Wk)  5 dV (x X -> Wv)
#include "enc_wgrad_job.h"

size_t enc_wgrad_part_floats(int64_t D, int64_t L) { return (size_t)L * EG_NMAT * WG_NSPLIT_MAX * D * D; }   // (sized for the most splits any width uses)

template <int D>
__global__ __launch_bounds__(512) void enc_wgrad_k(const float* __restrict__ tape, EncTape T, const float* __restrict__ gtape, int64_t NR,
                                                   const void* __restrict__ planp, int B, int S, int L, float* __restrict__ part,
                                                   const int64_t* __restrict__ seq, const float* __restrict__ contrib, float* __restrict__ ppart,
                                                   const float* __restrict__ dPtab) {
    extern __shared__ __align__(16) float lds[];
    if ((int)blockIdx.z == L) {
        if (!dPtab) return;
        wg_pos_job<D>(threadIdx.x, lds, blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y, B, S, seq, contrib, ppart);
        return;
    }
    const EncPlan PL = enc_plan_view(planp, B, S);
    wg_matrix_job<D>(threadIdx.x, lds, blockIdx.z, blockIdx.y, blockIdx.x, tape, T, gtape, NR, PL.hdr[1], part);
}

// optional: the dense Adam of every gradient element the reduction finishes (re_adam_fuse: arenas of one layout)
struct EncAdam {
    const float* gbase;
    float *p, *m, *v;
    const float* hyper;
    float b1, b2, omb1, omb2, eps, wd;
    const unsigned* gate;   // optional device word: non-zero = gradients are written, parameters and moments stay (a hand-over of this step timed out)
};
// What the end of an element's chain needs and does not depend on the partial sums -- the step scalars, the gate word, the element's parameter
// and moments -- is requested FIRST (eg_pre, unconditional loads), together with the partials: the launch is then one memory round trip deep
// instead of three (partials; scalars, on which a branch depended; parameter and moments).
struct EgPre {
    float ss, ib, p, m, v;
};
__device__ __forceinline__ EgPre eg_pre(const EncAdam& A, const float* d) {
    EgPre e{0.f, 0.f, 0.f, 0.f, 0.f};
    if (A.p) {
        const int64_t i = d - A.gbase;
        e.p = A.p[i]; e.m = A.m[i]; e.v = A.v[i];
        const unsigned gate_w = *(A.gate ? A.gate : reinterpret_cast<const unsigned*>(A.hyper));
        e.ss = A.hyper[0];
        e.ib = A.hyper[1];
        e.ib = (A.gate && gate_w != 0u) ? 0.f : e.ib;   // ({0, 0}: the caller gated this step off; gate: a hand-over of this step timed out)
    }
    return e;
}
__device__ __forceinline__ void eg_put(const EncAdam& A, float* d, float g, const EgPre& e) {
    *d = g;
    if (A.p && e.ib != 0.f) {
        const int64_t i = d - A.gbase;
        float pp = e.p, mm = e.m, vv = e.v;
        re_adam1(pp, mm, vv, g, A.b1, A.b2, A.omb1, A.omb2, e.ss, e.ib, A.eps, A.wd);     // (adam_vec4_dev's arithmetic)
        A.p[i] = pp; A.m[i] = mm; A.v[i] = vv;
    }
}

struct EncGradDst {
    float* p[SE_MAX_BLOCKS][14];  // per block: ABI order of the 12 block gradients, then g_last_w, g_last_b (last block only)
};

// blocks [0, nmat_blocks): 256 elements of the L * 6 * D * D weight gradients each (sum of the wg_nsplit(D) partials);
// the rest: 64 columns of one (block, vector) each, summed over the slabs of the workgroups that had work (4 waves x fixed order).
__global__ __launch_bounds__(256) void enc_grad_reduce_k(const float* __restrict__ part, const float* __restrict__ slab, int nwg,
                                                         const void* __restrict__ planp, int B, int S, int D, int L, EncGradDst dst,
                                                         int nmat_blocks, int nvec_blocks, const float* __restrict__ ppart, float inv_scale,
                                                         float* __restrict__ dPtab, int by_tile, EncAdam AD, unsigned* __restrict__ ticket) {
    re_kernarg_warm<re_kernarg_bytes(&enc_grad_reduce_k)>();
    const int tid = threadIdx.x;
    if (ticket && blockIdx.x == 0 && tid == 0) ticket[0] = 0u;   // (enc_tail_k's job counter: every job of this step has been taken)
    if ((int)blockIdx.x >= nmat_blocks + nvec_blocks) {
        // position-table gradient: the chunk partials of enc_wgrad_k in chunk order, / scale
        const int e = ((int)blockIdx.x - nmat_blocks - nvec_blocks) * 256 + tid;
        if (e >= S * D) return;
        const int p = e / D, cc = e % D, nch = (B + 63) / 64;
        const EgPre pre = eg_pre(AD, dPtab + e);
        float s = 0.f;
        for (int ch = 0; ch < nch; ++ch) s += ppart[((int64_t)p * nch + ch) * D + cc];
        eg_put(AD, dPtab + e, s * inv_scale, pre);
        return;
    }
    if ((int)blockIdx.x < nmat_blocks) {
        const int64_t e = (int64_t)blockIdx.x * 256 + tid;
        const int dd = D * D;
        if (e >= (int64_t)L * EG_NMAT * dd) return;
        const int lm = (int)(e / dd), off = (int)(e % dd);
        const int nsplit = wg_nsplit(D);
        const float* p = part + (int64_t)lm * nsplit * dd + off;
        const int l = lm / EG_NMAT, m = lm % EG_NMAT;
        float* const* P = dst.p[l];
        float* d = (m == 0) ? P[10] : (m == 1) ? P[8] : (m == 2) ? P[4] : P[2] + (m - 3) * dd;
        const EgPre pre = eg_pre(AD, d + off);
        float v[WG_NSPLIT_MAX];
#pragma unroll
        for (int i = 0; i < WG_NSPLIT_MAX; ++i) v[i] = p[(int64_t)(i < nsplit ? i : 0) * dd];   // (clamped, unconditional)
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WG_NSPLIT_MAX; ++i) s += i < nsplit ? v[i] : 0.f;                    // (x + 0 = x: the partials in split order)
        eg_put(AD, d + off, s, pre);
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
    float* dvec = nullptr;
    if (!(v >= 10 && l != L - 1)) {
        float* const* P = dst.p[l];
        switch (v) {
            case 0: case 1: case 2: dvec = P[3] + v * D; break;
            case 3: dvec = P[5]; break;
            case 4: dvec = P[9]; break;
            case 5: dvec = P[11]; break;
            case 6: dvec = P[0]; break;
            case 7: dvec = P[1]; break;
            case 8: dvec = P[6]; break;
            case 9: dvec = P[7]; break;
            case 10: dvec = P[12]; break;
            default: dvec = P[13]; break;
        }
        dvec += cg * 64 + lane;
    }
    const EgPre pre = (dvec && wave == 0) ? eg_pre(AD, dvec) : EgPre{0.f, 0.f, 0.f, 0.f, 0.f};
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
    if (!dvec) return;
    eg_put(AD, dvec, s, pre);
}

size_t enc_wgrad_ppart_floats(int64_t B, int64_t D) { return (size_t)64 * ((B + 63) / 64) * D; }

// The reduction launch behind the weight-gradient jobs (enc_wgrad_k here, or enc_tail_k's: enc_tail.hip -- `ticket` is its job counter).
int enc_grad_reduce_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* plan, const float* slab, int nwg, const float* part,
                           const float* ppart, float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b,
                           hipStream_t s, int by_tile, const re_adam_fuse* adam, unsigned* ticket, const unsigned* gate) {
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
                     (float)(1.0 - adam->beta2), (float)adam->eps, (float)adam->weight_decay, gate};
    }
    hipLaunchKernelGGL(enc_grad_reduce_k, dim3(nmat_blocks + nvec_blocks + npos_blocks), dim3(256), 0, s, part, slab, nwg, plan,
                       (int)B, (int)S, (int)D, (int)L, dst, nmat_blocks, nvec_blocks, ppart, emb_scale != 0.f ? 1.0f / emb_scale : 0.f,
                       dPtab, by_tile, AD, ticket);
    return re_launch_status();
}

int enc_wgrad_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* tape, const float* gtape, const void* plan, const float* slab,
                     int nwg, float* part, float* ppart, const int64_t* seq, const float* contrib, float emb_scale, float* dPtab,
                     float* const* block_grads, float* g_last_w, float* g_last_b, hipStream_t s, int by_tile, const re_adam_fuse* adam) {
    if (D != 64 && D != 128) return RE_EUNSUPPORTED;
    const EncTape T = enc_tape_layout(B, S, D, L);
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    const dim3 grid((unsigned)wg_nsplit((int)D), EG_NMAT, (unsigned)(L + (dPtab ? 1 : 0)));
    if (D == 128) {
        using C = EC<128>;
        const size_t ldsb = (size_t)wg_job_lds_floats<128>() * sizeof(float);
        if (hipFuncSetAttribute((const void*)enc_wgrad_k<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(enc_wgrad_k<128>, grid, dim3(C::NT), ldsb, s, (const float*)tape, T, gtape, NR, plan, (int)B, (int)S, (int)L, part, seq, contrib,
                           ppart, (const float*)dPtab);
    } else {
        using C = EC<64>;
        const size_t ldsb = (size_t)wg_job_lds_floats<64>() * sizeof(float);
        hipLaunchKernelGGL(enc_wgrad_k<64>, grid, dim3(C::NT), ldsb, s, (const float*)tape, T, gtape, NR, plan, (int)B, (int)S, (int)L, part, seq, contrib,
                           ppart, (const float*)dPtab);
    }
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    return enc_grad_reduce_launch(B, S, D, L, plan, slab, nwg, part, ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, s, by_tile, adam, nullptr, nullptr);
}
