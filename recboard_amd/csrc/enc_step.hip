// K6/K7 training step of the encoder as ONE launch: per work item the forward (enc_fwd_item.h, loss head included) and,
// straight after it, the backward (enc_bwd_item.h) -- an item's gradient chain needs nothing from any other item (the weight
// gradients, which do, stay in enc_wgrad.hip).  Against the two launches: no launch boundary in the middle of the step, the
// backward's first requests overlap the forward's tail, and the tape is read back by the workgroup that wrote it a moment ago
// (same XCD: L2 hits).  The results are those of re_sasrec_encoder_fwd_loss followed by re_sasrec_encoder_bwd, bit for bit.
#include <math.h>

#include "enc_fwd_item.h"
#include "enc_tile_prep.h"
#include "enc_bwd_item.h"

template <int D>
__global__ __launch_bounds__(512) void enc_step_k(SeEmbed em, const int64_t* __restrict__ seq, int B, int S, int L, SasrecParams P,
                                                  float drop_scale, uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                  float* __restrict__ tape, EncTape T, const void* __restrict__ planp, EncHead H,
                                                  float* __restrict__ dOut, float* __restrict__ gtape, float* __restrict__ slab,
                                                  const uint32_t* __restrict__ seed_dev, float emb_scale) {
    if (seed_dev) seed ^= seed_dev[0];
    extern __shared__ __align__(16) float lds[];
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_items = PL.hdr[0];
    if (PL.hdr[7] == 1) return;   // (the one-tile-per-workgroup kernel in front of this launch has run the step: enc_tile.hip)
    if (enc_split_plan_rejected(PL, tape + T.off_FLAGS, enc_plan_max_tiles(B, S) * EP_FLAG_WORDS, H.loss)) return;
    using C = EC<D>;
    __shared__ int h_gid[C::ROWS], h_first[C::ROWS], h_pad[C::ROWS], h_sid[C::ROWS], h_start[C::ROWS];
    __shared__ float h_mean[C::ROWS], h_rstd[C::ROWS];
    const EncHandoff HO{h_gid, h_first, h_pad, h_sid, h_start, h_mean, h_rstd, lds + 4 * C::BUF};   // (the fifth tile buffer is free at both ends)
    for (int k = 0; k * (int)gridDim.x < n_items; ++k) {
        const int wi = enc_item_of(k, blockIdx.x, gridDim.x);
        if (wi >= n_items) continue;
        enc_fwd_item<D, true, true>(nullptr, em, seq, B, S, L, P, drop_scale, thresh, seed, u, tape, T, PL, 0, H, lds, wi, k, &HO);
        re_sync_full();    // (a full barrier: the item's tape and upstream-gradient rows are written before they are read back)
        enc_bwd_item<D>(H.dU_rows, seq, B, S, L, P, drop_scale, thresh, seed, tape, T, PL, dOut, gtape, slab, 1, emb_scale, 1,
                        H.g_rows, lds, wi, k, &HO);
        __syncthreads();
    }
}

// ---- weight gradients (enc_wgrad.hip) ---------------------------------------------------------------------------------------
int enc_wgrad_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* tape, const float* gtape, const void* plan, const float* slab,
                     int nwg, float* part, float* ppart, const int64_t* seq, const float* contrib, float emb_scale, float* dPtab,
                     float* const* block_grads, float* g_last_w, float* g_last_b, hipStream_t s, int by_tile = 0, const re_adam_fuse* adam = nullptr);
size_t enc_wgrad_part_floats(int64_t D, int64_t L);
size_t enc_wgrad_ppart_floats(int64_t B, int64_t D);
extern "C" size_t re_sasrec_encoder_bwd_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t L);
// ---- the one-tile-per-workgroup form of the step (enc_tile.hip: four waves per tile at D = 64, eight at D = 128)
int enc_tile_step_launch(int64_t D, const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P, float ds, uint32_t thresh,
                         uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan, int grid, const EncHead& H, float* dx0,
                         float* gtape, float* slab, float scale, uint32_t* wf, float* xch, int prep, hipStream_t s);

template <int D>
static int enc_step_launch_d(const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P, float ds,
                             uint32_t thresh, uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan, int grid,
                             const EncHead& H, float* dx0, float* gtape, float* slab, float scale, hipStream_t s) {
    using C = EC<D>;
    const EncTape T = enc_tape_layout(B, S, D, L);
    const size_t lf = (size_t)(5 * C::BUF + C::PBUF + 2 * C::PRE);
    const size_t lb = (size_t)(5 * C::BUF + 2 * C::PBUF + 2 * C::PRE);
    const size_t ldsb = (lf > lb ? lf : lb) * sizeof(float);
    auto k = enc_step_k<D>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), ldsb, s, em, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)tape, T, plan, H, dx0,
                       gtape, slab, seed_dev, scale);
    return hipGetLastError() == hipSuccess ? RE_OK : RE_ELAUNCH;
}

// part: a mask of what to launch -- 1 = the one-tile-per-workgroup kernels (D = 64; runs when the plan allows it), 2 = the workgroup-per-item
// kernel (D = 128 always; D = 64 when the plan does not allow the tile kernel: exactly one of the two does the work, the other returns at
// once), 4 = the weight gradients (enc_wgrad_k + enc_grad_reduce_k, from the tape the item kernels left); 0 = 7 = the whole step.
// + 8: the tile kernel's weight fragments were prepared by re_sasrec_batch_prep_w for this step (no enc_tile_prep_k launch).
// A caller with other work depending on the item kernels alone (the item table's scatter-add) runs the branches on two streams.
// the device's compute units (a constant of the device, asked per launch: no state kept)
static int enc_device_cus() {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) return 256;
    return n;
}

extern "C" int re_sasrec_encoder_step_part(const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, const int64_t* pos,
                                           const int64_t* neg, int64_t B, int64_t S, int64_t D, int64_t L, const float* const* block_params,
                                           const float* last_w, const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                                           const void* plan, int32_t ncu, float* u, void* tape, size_t tape_bytes, int64_t e_off, int kind,
                                           const int32_t* count, float* loss, float* dU_rows, float* g_rows, int32_t* keys, void* loss_ws,
                                           size_t loss_ws_bytes, float* dx0, float* dPtab, float* const* block_grads, float* g_last_w,
                                           float* g_last_b, void* ws, size_t ws_bytes, int32_t part, const re_adam_fuse* adam, re_stream_t stream) {
    re_clear_error();
    if (part < 0 || part > 15) return RE_EINVAL;
    const bool frag_ready = (part & 8) != 0;
    part &= 7;
    if (part == 0) part = 7;
    if (B == 0) return RE_OK;
    if (!seq || !pos || !neg || !u || !plan || !tape || !E || !Ptab || !count || !loss || !dU_rows || !g_rows || !keys || !loss_ws || !dx0 ||
        !dPtab || !block_params || !block_grads || !g_last_w || !g_last_b || !last_w || !last_b || !ws || B < 0 || R <= 0)
        return RE_EINVAL;
    if (kind != RE_LOSS_BCE && kind != RE_LOSS_BPR) return RE_EINVAL;
    if ((reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(Ptab) | reinterpret_cast<uintptr_t>(dU_rows) | reinterpret_cast<uintptr_t>(g_rows)) & 15u)
        return RE_EUNSUPPORTED;
    if ((D != 64 && D != 128) || S < 1 || S > 64 || L < 1 || L > SE_MAX_BLOCKS || R >= 0x7FFFFFFFll) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    if (loss_ws_bytes < 256 || ws_bytes < re_sasrec_encoder_bwd_workspace_bytes(B, S, D, L)) return RE_EWORKSPACE;
    for (int64_t i = 0; i < 12 * L; ++i)
        if (!block_grads[i]) return RE_EINVAL;
    SasrecParams P;
    if (!se_fill_params(P, block_params, L, last_w, last_b)) return RE_EINVAL;
    if (tape_bytes < (size_t)enc_tape_layout(B, S, D, L).total * sizeof(float)) return RE_EWORKSPACE;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const SeEmbed em{E, Ptab, R, scale};
    const EncHead H{E, R, e_off, pos, neg, kind, count, loss, dU_rows, g_rows, keys, (unsigned long long*)loss_ws};
    if (ncu < 1) ncu = 256;
    const int64_t mt = enc_plan_max_tiles(B, S);
    const int grid = (int)(mt < ncu ? mt : ncu);
    if (grid > 1024) return RE_EUNSUPPORTED;
    const EncBwdWs Wk = enc_bwd_ws(ws, B, S, D, L);
    float *slab = Wk.slab, *wpart = Wk.wpart, *ppart = Wk.ppart, *gtape = Wk.gtape;
    hipStream_t s = (hipStream_t)stream;
    {
        // one tile per workgroup (enc_tile.hip) when the plan says every tile can have a resident workgroup (hdr[7]); the workgroup-per-item
        // kernel is launched behind it and returns at once in that case -- the plan lives in device memory, so both are always enqueued
        uint32_t* wf = Wk.wf;
        float* xch = Wk.xch;
        // (looped form: the resident workgroups, one per CU, further tiles from a counter; else a workgroup per tile, <= 1024 tiles by the plan's rule)
        // the looped form's grid = the RESIDENT workgroups (block b owns tile b; the rest come from a counter): CUs x workgroups per CU, the CU
        // count clamped to the device's (a caller's larger `ncu` would start blocks that cannot be resident while their partners spin)
        const int64_t res = (int64_t)(ncu < enc_device_cus() ? ncu : enc_device_cus()) * enc_tile_wg_per_cu(D);
        const int tgrid = enc_tile_looped(B, S) ? (int)(mt < res ? mt : res) : (int)(mt < 1024 ? mt : 1024);
        const int wgrid = (int)(mt < ncu ? mt : ncu);
        if (part & 1) {
            const int rcw = enc_tile_step_launch(D, em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, tgrid, H, dx0, gtape, slab, scale, wf, xch,
                                                 frag_ready ? 0 : 1, s);
            if (rcw != RE_OK) return rcw;
        }
        if (part & 2) {
            const int rco = D == 128 ? enc_step_launch_d<128>(em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, wgrid, H, dx0, gtape, slab, scale, s)
                                     : enc_step_launch_d<64>(em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, wgrid, H, dx0, gtape, slab, scale, s);
            if (rco != RE_OK) return rco;
        }
        if (!(part & 4)) return RE_OK;
        return enc_wgrad_launch(B, S, D, L, tape, gtape, plan, slab, wgrid, wpart, ppart, seq, dx0, scale, dPtab, block_grads, g_last_w, g_last_b, s, 1, adam);
    }
}

extern "C" int re_sasrec_encoder_step(const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, const int64_t* pos,
                                      const int64_t* neg, int64_t B, int64_t S, int64_t D, int64_t L, const float* const* block_params,
                                      const float* last_w, const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                                      const void* plan, int32_t ncu, float* u, void* tape, size_t tape_bytes, int64_t e_off, int kind,
                                      const int32_t* count, float* loss, float* dU_rows, float* g_rows, int32_t* keys, void* loss_ws,
                                      size_t loss_ws_bytes, float* dx0, float* dPtab, float* const* block_grads, float* g_last_w,
                                      float* g_last_b, void* ws, size_t ws_bytes, re_stream_t stream) {
    return re_sasrec_encoder_step_part(E, R, Ptab, scale, seq, pos, neg, B, S, D, L, block_params, last_w, last_b, drop_p, seed, seed_dev, plan, ncu, u,
                                       tape, tape_bytes, e_off, kind, count, loss, dU_rows, g_rows, keys, loss_ws, loss_ws_bytes, dx0, dPtab,
                                       block_grads, g_last_w, g_last_b, ws, ws_bytes, 0, nullptr, stream);
}

// Resident workgroups per CU the tile kernels of THIS library are built for (1; 2 in the experiment builds): the batch plan's residency rule
// (split_long & 8) has to count the same number, so the host asks instead of assuming.
extern "C" int re_tile_wgs_per_cu(int64_t D) { return enc_tile_wg_per_cu(D); }

// 1: a plan made for this shape with split_long = 4 (+ 8) -- no splitting, the tile kernels forced -- hands EVERY batch to the tile kernels
// (hdr[7] = 1 whatever the batch: the looped form takes any number of tiles up to the inbox cap, and tickets in tile order cannot deadlock a
// chain), so the caller may leave the workgroup-per-item launch out of the step (part & 2: an empty launch costs 5 us of a 90 us step).
extern "C" int re_sasrec_tile_step_certain(int64_t B, int64_t S, int64_t D) {
    return (D == 64 && B > 0 && S > 0 && S <= 64 && enc_tile_looped(B, S) && enc_plan_max_tiles(B, S) <= ENC_XCH_TILE_CAP) ? 1 : 0;
}
