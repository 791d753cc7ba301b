// K6/K7 backward (kernel + entry point; the item code is enc_bwd_item.h): ONE launch for all blocks (l = L-1 .. 0), one work item per workgroup iteration, everything in LDS.
//
// Gradient of the encoder of enc_fwd.hip (SASRec/main.py:163-176 + :31-50 + lastLN) w.r.t. its input rows, reading the forward's
// tape (x, q, k, v, P, o, x1, relu(h), LN statistics) instead of recomputing the forward; dropout masks are regenerated from
// (seed, stream, index).  An item's gradient chain is independent of every other item's, so the block loop runs inside the
// kernel and dX never leaves LDS between blocks.
//
// What this kernel does NOT do is the six weight gradients dW = dY^T X per block: those are contractions over ALL rows of the
// batch.  The kernel writes its six dY operands per block (dO2, dH, dX1, dQ, dK, dV) to a gradient tape, and
// enc_wgrad.hip computes the weight gradients as split-K products over the compact rows at full-chip parallelism -- instead
// of six more MFMA phases on every item's critical path plus one D x D slab per workgroup per matrix to reduce.
// Bias / LayerNorm gradients are column sums: per-thread partials, one small slab per workgroup and block.
//
// MFMA-bound work: 10 products of [16 nt] x D x D per block per item.
#include <math.h>

#include "enc_common.h"

#ifdef ENC_PROFILE
__device__ unsigned long long g_bwd_marks[ENC_MARKS];
extern "C" int re_dbg_enc_marks_bwd(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bwd_marks), sizeof(unsigned long long) * ENC_MARKS) == hipSuccess ? 0 : 1;
}
#endif

#include "enc_bwd_item.h"
#include "enc_tile_prep.h"

template <int D>
__global__ __launch_bounds__(512) void enc_bwd_k(const float* __restrict__ dIn, const int64_t* __restrict__ seq, int B, int S, int L,
                                                 SasrecParams P, float drop_scale, uint32_t thresh, uint32_t seed,
                                                 const float* __restrict__ tape, EncTape T, const void* __restrict__ planp,
                                                 float* __restrict__ dOut, float* __restrict__ gtape, float* __restrict__ slab,
                                                 const uint32_t* __restrict__ seed_dev, int fuse_embed, float emb_scale, int in_rows,
                                                 float* __restrict__ dOutRows) {
    if (seed_dev) seed ^= seed_dev[0];   // per-step seed kept in device memory (hipGraph replays)
    extern __shared__ __align__(16) float lds[];
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_items = PL.hdr[0];
    if (enc_split_plan_rejected(PL, const_cast<float*>(tape) + T.off_FLAGS, enc_plan_max_tiles(B, S) * EP_FLAG_WORDS, nullptr)) return;
    for (int k = 0; k * (int)gridDim.x < n_items; ++k) {
        const int wi = enc_item_of(k, blockIdx.x, gridDim.x);
        if (wi >= n_items) continue;
        enc_bwd_item<D>(dIn, seq, B, S, L, P, drop_scale, thresh, seed, tape, T, PL, dOut, gtape, slab, fuse_embed, emb_scale, in_rows, dOutRows,
                        lds, wi, k);
    }
}

// ---- weight gradients (enc_wgrad.hip) ---------------------------------------------------------------------------------------
int enc_wgrad_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* tape, const float* gtape, const void* plan, const float* slab,
                     int nwg, float* part, float* ppart, const int64_t* seq, const float* contrib, float emb_scale, float* dPtab,
                     float* const* block_grads, float* g_last_w, float* g_last_b, hipStream_t s, int by_tile = 0, const re_adam_fuse* adam = nullptr);
size_t enc_wgrad_part_floats(int64_t D, int64_t L);
size_t enc_wgrad_ppart_floats(int64_t B, int64_t D);

static inline int enc_bwd_grid(int64_t B, int64_t S, int32_t ncu) {
    const int64_t mt = enc_plan_max_tiles(B, S);
    if (ncu < 1) ncu = 256;
    return (int)(mt < ncu ? mt : ncu);
}

extern "C" size_t re_sasrec_encoder_bwd_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t L) {
    if (B <= 0 || S <= 0 || D <= 0 || L <= 0) return 256;
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    const int64_t nwg = enc_slab_rows(B, S);   // upper bound of the launch grid (ncu), or one row per tile
    // (+ the weight-fragment planes of the wave-per-tile step, enc_wave.hip: L x 6 matrices x 2 orientations x 16 KB, 256-byte aligned)
    return (size_t)(nwg * L * EG_NVEC * D + enc_wgrad_part_floats(D, L) + enc_wgrad_ppart_floats(B, D) + L * EG_NMAT * NR * D) * sizeof(float) + 512 +
           enc_tile_wfrag_bytes(L, D) + 512 + enc_tile_xch_bytes(B, S, D, L) + 256;   // (+ the tiles' dK / dV inboxes)
}

// Where enc_bwd_ws (enc_tile_prep.h) puts the regions of that workspace for a buffer at address `base` (not dereferenced):
// out[0..6] = byte offsets of slab, matrix partials, position partials, gradient tape, weight fragments, inboxes, end.
extern "C" int re_sasrec_encoder_bwd_workspace_layout(int64_t B, int64_t S, int64_t D, int64_t L, uint64_t base, uint64_t* out) {
    if (!out || B <= 0 || S <= 0 || S > 64 || (D != 64 && D != 128) || L < 1 || L > SE_MAX_BLOCKS) return RE_EINVAL;
    const EncBwdWs W = enc_bwd_ws((void*)(uintptr_t)base, B, S, D, L);
    const char* b = (const char*)(uintptr_t)base;
    out[0] = (uint64_t)((const char*)W.slab - b); out[1] = (uint64_t)((const char*)W.wpart - b); out[2] = (uint64_t)((const char*)W.ppart - b);
    out[3] = (uint64_t)((const char*)W.gtape - b); out[4] = (uint64_t)((const char*)W.wf - b); out[5] = (uint64_t)((const char*)W.xch - b);
    out[6] = (uint64_t)W.bytes;
    return RE_OK;
}

// dPtab == NULL: dx0 [B,S,D] receives the gradient w.r.t. x0 (rows of real tokens only).  Otherwise re_sasrec_embed_bwd is fused in:
// dx0 receives the item-gradient contribution rows (pad mask, embedding dropout mask, * scale) and dPtab [S, D] the position-table gradient.
extern "C" int re_sasrec_encoder_bwd(const float* dU, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                                     const float* const* block_params, const float* last_w, const float* last_b, float drop_p, uint32_t seed,
                                     const uint32_t* seed_dev, const void* tape, const void* plan, int32_t ncu, float scale, float* dx0,
                                     float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b, const float* dU_rows,
                                     float* dx0_rows, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (dU_rows) dU = dU_rows;
    if (!dU || !seq || !tape || !plan || !dx0 || !block_params || !block_grads || !g_last_w || !g_last_b || !last_w || !last_b || !ws || B < 0)
        return RE_EINVAL;
    if ((D != 64 && D != 128) || S < 1 || S > 64 || L < 1 || L > SE_MAX_BLOCKS) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    if (ws_bytes < re_sasrec_encoder_bwd_workspace_bytes(B, S, D, L)) return RE_EWORKSPACE;
    for (int64_t i = 0; i < 12 * L; ++i)
        if (!block_grads[i]) return RE_EINVAL;
    SasrecParams P;
    if (!se_fill_params(P, block_params, L, last_w, last_b)) return RE_EINVAL;
    const EncTape T = enc_tape_layout(B, S, D, L);
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int grid = enc_bwd_grid(B, S, ncu);
    if (grid > 1024) return RE_EUNSUPPORTED;
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    const EncBwdWs Wk = enc_bwd_ws(ws, B, S, D, L);
    float *slab = Wk.slab, *part = Wk.wpart, *ppart = Wk.ppart, *gtape = Wk.gtape;
    hipStream_t s = (hipStream_t)stream;
    if (D == 128) {
        using C = EC<128>;
        const size_t ldsb = (size_t)(5 * C::BUF + 2 * C::PBUF + 2 * C::PRE) * sizeof(float);
        auto kf = enc_bwd_k<128>;
        if (hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(kf, dim3(grid), dim3(C::NT), ldsb, s, dU, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, (const float*)tape, T, plan, dx0,
                           gtape, slab, seed_dev, dPtab ? 1 : 0, scale, dU_rows ? 1 : 0, dx0_rows);
    } else {
        using C = EC<64>;
        const size_t ldsb = (size_t)(5 * C::BUF + 2 * C::PBUF + 2 * C::PRE) * sizeof(float);
        auto kf = enc_bwd_k<64>;
        if (hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(kf, dim3(grid), dim3(C::NT), ldsb, s, dU, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, (const float*)tape, T, plan, dx0,
                           gtape, slab, seed_dev, dPtab ? 1 : 0, scale, dU_rows ? 1 : 0, dx0_rows);
    }
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    (void)NR;
    return enc_wgrad_launch(B, S, D, L, tape, gtape, plan, slab, grid, part, ppart, seq, dx0, scale, dPtab, block_grads, g_last_w, g_last_b, s);
}
