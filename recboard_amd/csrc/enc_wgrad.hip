// Weight gradients of the SASRec encoder, after enc_bwd.hip: dW = dY^T X for the six linear maps of every block, as split-K
// products over the batch's compact rows (the contraction runs over ALL rows, ~3 600 on a Beauty-shaped batch): grid =
// (row splits) x (6 matrices) x (blocks), every workgroup accumulates one D x D partial over its share of the row tiles
// (operands are contiguous row ranges of the tape / gradient tape, staged through LDS), `enc_grad_reduce_k` adds the partials
// and the workgroups' vector-gradient slabs in a fixed order (deterministic, no float atomics).  The position-table gradient
// (sum over the batch of the embedding-backward rows at every position, SASRec/main.py:183 `+ Position(positions)`) rides in
// the same launch as one more grid plane.
//   gradient-tape order: 0 dO2 (x HR -> W2)  1 dH (x Y -> W1)  2 dX1 (x O -> Wo)  3 dQ (x A -> Wq)  4 dK (x X -> Wk)  5 dV (x X -> Wv)
#include "enc_grad_reduce.h"

size_t enc_wgrad_part_floats(int64_t D, int64_t L) { return (size_t)L * EG_NMAT * WG_NSPLIT_MAX * D * D; }   // (sized for the most splits any width uses)

template <int D>
__global__ __launch_bounds__(512) void enc_wgrad_k(const float* __restrict__ tape, EncTape T, const float* __restrict__ gtape, int64_t NR,
                                                   const void* __restrict__ planp, int B, int S, int L, float* __restrict__ part,
                                                   const int64_t* __restrict__ seq, const float* __restrict__ contrib, float* __restrict__ ppart,
                                                   const float* __restrict__ dPtab) {
    extern __shared__ __align__(16) float lds[];
    if ((int)blockIdx.z == L) {
        if (!dPtab) return;
        for (int g = blockIdx.y * gridDim.x + blockIdx.x; g < WG_POS_GROUPS; g += gridDim.x * gridDim.y)   // (144 workgroups at D = 64, 72 at 128)
            wg_pos_job<D>(threadIdx.x, lds, g, B, S, seq, contrib, ppart);
        return;
    }
    const EncPlan PL = enc_plan_view(planp, B, S);
    wg_matrix_job<D>(threadIdx.x, lds, blockIdx.z, blockIdx.y, blockIdx.x, tape, T, gtape, NR, PL.hdr[1], part);
}

// blocks [0, nmat_blocks): 256 elements of the L * 6 * D * D weight gradients each (sum of the wg_nsplit(D) partials);
// the rest: 64 columns of one (block, vector) each, summed over the slabs of the workgroups that had work (4 waves x fixed order); then the
// position table's blocks (enc_grad_reduce.h holds the bodies: the step tail's ticket queue runs the same ones as its last jobs).
__global__ __launch_bounds__(256) void enc_grad_reduce_k(EgReduce R, unsigned* __restrict__ ticket) {
    re_kernarg_warm<re_kernarg_bytes(&enc_grad_reduce_k)>();
    const int tid = threadIdx.x;
    if (ticket && blockIdx.x == 0 && tid == 0) ticket[0] = 0u;   // (enc_tail_k's job counter: every job of this step has been taken)
    __shared__ float red[4][64];
    if ((int)blockIdx.x >= R.nmat_blocks + R.nvec_blocks) {
        eg_reduce_pos(R, (int)blockIdx.x - R.nmat_blocks - R.nvec_blocks, tid, red);
        return;
    }
    if ((int)blockIdx.x < R.nmat_blocks) {
        eg_reduce_mat(R, (int)blockIdx.x, tid);
        return;
    }
    const EgVec V = eg_reduce_vec_a(R, (int)blockIdx.x - R.nmat_blocks, tid, red);
    __syncthreads();
    eg_reduce_vec_b(R, V, tid, red);
}

size_t enc_wgrad_ppart_floats(int64_t, int64_t D) { return (size_t)64 * WG_POS_GROUPS * D; }   // [S <= 64][groups][D]

// the reduction's arguments (shared by the launch below and the step tail's queue, enc_tail.hip)
int enc_grad_reduce_args(EgReduce& R, int64_t B, int64_t S, int64_t D, int64_t L, const void* plan, const float* slab, int nwg, const float* part,
                         const float* ppart, float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b, int by_tile,
                         const re_adam_fuse* adam, const unsigned* gate, int nsplit) {
    R = EgReduce{};
    for (int64_t l = 0; l < SE_MAX_BLOCKS; ++l)
        for (int i = 0; i < 14; ++i) R.dst.p[l][i] = (l < L && i < 12) ? block_grads[12 * l + i] : (i == 12 ? g_last_w : g_last_b);
    R.nmat_blocks = (int)((L * EG_NMAT * D * D + 255) / 256);
    R.nvec_blocks = (int)(L * EG_NVEC * (D / 64));
    R.npos_blocks = dPtab ? (int)((S * D + 63) / 64) : 0;
    if (adam) {
        if (!adam->grad_base || !adam->param || !adam->m || !adam->v || !adam->hyper) return RE_EINVAL;
        R.AD = EncAdam{adam->grad_base, adam->param, adam->m, adam->v, adam->hyper, (float)adam->beta1, (float)adam->beta2, (float)(1.0 - adam->beta1),
                       (float)(1.0 - adam->beta2), (float)adam->eps, (float)adam->weight_decay, gate};
    }
    R.part = part; R.slab = slab; R.nwg = nwg; R.planp = plan; R.B = (int)B; R.S = (int)S; R.D = (int)D; R.L = (int)L;
    R.nsplit = nsplit > 0 ? nsplit : wg_nsplit((int)D);
    R.ppart = ppart; R.inv_scale = emb_scale != 0.f ? 1.0f / emb_scale : 0.f; R.dPtab = dPtab; R.by_tile = by_tile;
    return RE_OK;
}

// The reduction launch behind the weight-gradient jobs (enc_wgrad_k here, or enc_tail_k's: enc_tail.hip -- `ticket` is its job counter).
int enc_grad_reduce_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* plan, const float* slab, int nwg, const float* part,
                           const float* ppart, float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b,
                           hipStream_t s, int by_tile, const re_adam_fuse* adam, unsigned* ticket, const unsigned* gate, int nsplit) {
    EgReduce R;
    const int rc = enc_grad_reduce_args(R, B, S, D, L, plan, slab, nwg, part, ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, by_tile, adam, gate, nsplit);
    if (rc != RE_OK) return rc;
    hipLaunchKernelGGL(enc_grad_reduce_k, dim3(R.nmat_blocks + R.nvec_blocks + R.npos_blocks), dim3(256), 0, s, R, ticket);
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
    return enc_grad_reduce_launch(B, S, D, L, plan, slab, nwg, part, ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, s, by_tile, adam, nullptr, nullptr, 0);
}
