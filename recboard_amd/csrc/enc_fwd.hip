// K6/K7 forward (kernel + entry point; the item code is enc_fwd_item.h): the whole SASRec encoder (embedding front end, all blocks, lastLN) for one work item per workgroup,
// activations in LDS (enc_common.h: real-token rows only, 1..MAXT tiles of 16 rows per item).
//
// Reference restated: SASRec/main.py:178-193 (encode), :163-176 (after_one_block), :31-50 (PointWiseFeedForward).
//   x0 = seq == 0 ? 0 : dropout(E[seq] * sqrt(D) + P[s])
//   q = LN_a(x) Wq^T + bq;  k = x Wk^T + bk;  v = x Wv^T + bv          (K,V are NOT layer-normed)
//   A = dropout(softmax(q k^T / sqrt(D) + causal));  x1 = (A v) Wo^T + bo + x   (pad positions ARE attended as keys)
//   y = LN_f(x1);  x' = dropout2(relu(dropout1(y W1^T + b1)) W2^T + b2) + y;  x'[pad] = 0
//   u = LN_last(x_L)
// Weights stream from L2 as MFMA B fragments straight into registers (each of the NS column-strip waves reads its own 16
// output rows of W: no redundancy worth an LDS staging pass); in training every intermediate the backward and the
// weight-gradient kernel need is written once to the tape (contiguous per item).
// MFMA-bound work: 8 products of [16 nt] x D x D per block per item.
#include <math.h>

#include "enc_common.h"

#ifdef ENC_PROFILE
__device__ unsigned long long g_fwd_marks[ENC_MARKS];
extern "C" int re_dbg_enc_marks_fwd(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fwd_marks), sizeof(unsigned long long) * ENC_MARKS) == hipSuccess ? 0 : 1;
}
#endif

#include "enc_fwd_item.h"

template <int D, bool TRAIN>
__global__ __launch_bounds__(512) void enc_fwd_k(const float* __restrict__ x0, SeEmbed em, const int64_t* __restrict__ seq, int B, int S, int L,
                                                 SasrecParams P, float drop_scale, uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                 float* __restrict__ tape, EncTape T, const void* __restrict__ planp, int fill_pads,
                                                 const uint32_t* __restrict__ seed_dev) {
    if (seed_dev) seed ^= seed_dev[0];      // per-step seed kept in device memory (hipGraph replays)
    extern __shared__ __align__(16) float lds[];
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_items = PL.hdr[0];
    if (TRAIN && enc_split_plan_rejected(PL, tape + T.off_FLAGS, enc_plan_max_tiles(B, S) * EP_FLAG_WORDS, nullptr)) return;
    const EncHead H{};
    for (int k = 0; k * (int)gridDim.x < n_items; ++k) {
        const int wi = enc_item_of(k, blockIdx.x, gridDim.x);
        if (wi >= n_items) continue;
        enc_fwd_item<D, TRAIN, false>(x0, em, seq, B, S, L, P, drop_scale, thresh, seed, u, tape, T, PL, fill_pads, H, lds, wi, k);
    }
}

// the training forward with the loss head folded into the item's tail (one launch less, and the criterion's gathers overlap the last block)
template <int D>
__global__ __launch_bounds__(512) void enc_fwd_loss_k(SeEmbed em, const int64_t* __restrict__ seq, int B, int S, int L, SasrecParams P,
                                                      float drop_scale, uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                      float* __restrict__ tape, EncTape T, const void* __restrict__ planp, EncHead H,
                                                      const uint32_t* __restrict__ seed_dev) {
    if (seed_dev) seed ^= seed_dev[0];
    extern __shared__ __align__(16) float lds[];
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_items = PL.hdr[0];
    if (enc_split_plan_rejected(PL, tape + T.off_FLAGS, enc_plan_max_tiles(B, S) * EP_FLAG_WORDS, H.loss)) return;
    for (int k = 0; k * (int)gridDim.x < n_items; ++k) {
        const int wi = enc_item_of(k, blockIdx.x, gridDim.x);
        if (wi >= n_items) continue;
        enc_fwd_item<D, true, true>(nullptr, em, seq, B, S, L, P, drop_scale, thresh, seed, u, tape, T, PL, 0, H, lds, wi, k);
    }
}

extern "C" size_t re_sasrec_tape_bytes(int64_t B, int64_t S, int64_t D, int64_t L) {
    if (B <= 0 || S <= 0 || D <= 0 || L <= 0) return 256;
    return (size_t)enc_tape_layout(B, S, D, L).total * sizeof(float);
}

// The tape's array offsets (floats), for tools and tests that read a tape back: out[0..n) in this order --
//   per_block, X, A, Q, K, V, O, X1, Y, HR, P, SA, SF, PP, MK, XL, SL, FLAGS, total  (19 values; arrays of block l start at l * per_block)
extern "C" int re_sasrec_tape_layout(int64_t B, int64_t S, int64_t D, int64_t L, int64_t* out, int64_t n) {
    if (B <= 0 || S <= 0 || D <= 0 || L <= 0 || !out || n < 0) return RE_EINVAL;
    const EncTape t = enc_tape_layout(B, S, D, L);
    const int64_t v[19] = {t.per_block, t.off_X, t.off_A, t.off_Q, t.off_K, t.off_V, t.off_O, t.off_X1, t.off_Y, t.off_HR, t.off_P, t.off_SA,
                           t.off_SF, t.off_PP, t.off_MK, t.off_XL, t.off_SL, t.off_FLAGS, t.total};
    for (int64_t i = 0; i < n && i < 19; ++i) out[i] = v[i];
    return RE_OK;
}

template <int D>
static int enc_fwd_launch_d(const float* x0, const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P,
                            float ds, uint32_t thresh, uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan,
                            int grid, int fill_pads, hipStream_t s) {
    using C = EC<D>;
    const EncTape T = enc_tape_layout(B, S, D, L);
    const size_t ldsb = (size_t)(5 * C::BUF + C::PBUF + 2 * C::PRE) * sizeof(float);
    if (tape) {
        auto k = enc_fwd_k<D, true>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), ldsb, s, x0, em, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)tape, T, plan,
                           fill_pads, seed_dev);
    } else {
        auto k = enc_fwd_k<D, false>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), ldsb, s, x0, em, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)nullptr, T, plan,
                           fill_pads, seed_dev);
    }
    return re_launch_status();
}

// x0 == NULL: the input rows are built from (E, R, P, scale) inside the kernel (re_sasrec_embed fused in); otherwise x0 [B,S,D] is given.
extern "C" int re_sasrec_encoder_fwd(const float* x0, const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, int64_t B,
                                     int64_t S, int64_t D, int64_t L, const float* const* block_params, const float* last_w,
                                     const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev, const void* plan, int32_t ncu,
                                     float* u, void* tape, size_t tape_bytes, int32_t fill_pads, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!seq || !u || !plan || B < 0) return RE_EINVAL;
    if (!x0 && (!E || !Ptab || R <= 0)) return RE_EINVAL;
    if (!x0 && ((reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(Ptab)) & 15u)) return RE_EUNSUPPORTED;
    if ((D != 64 && D != 128) || S < 1 || S > 64 || L > SE_MAX_BLOCKS) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    SasrecParams P;
    if (!se_fill_params(P, block_params, L, last_w, last_b)) return RE_EINVAL;
    if (tape && tape_bytes < (size_t)enc_tape_layout(B, S, D, L).total * sizeof(float)) return RE_EWORKSPACE;
    if (D == 128 && !tape && S > 16 * EC<128>::MAXT) return RE_EINVAL;   // parts of a long sequence hand k, v over through the tape
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const SeEmbed em{x0 ? nullptr : E, Ptab, R, scale};
    if (ncu < 1) ncu = 256;
    const int64_t mt = enc_plan_max_tiles(B, S);
    const int grid = (int)(mt < ncu ? mt : ncu);   // one resident workgroup per CU; items beyond the grid are taken in further rounds
    if (D == 128) return enc_fwd_launch_d<128>(x0, em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, grid, fill_pads, (hipStream_t)stream);
    return enc_fwd_launch_d<64>(x0, em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, grid, fill_pads, (hipStream_t)stream);
}

template <int D>
static int enc_fwd_loss_launch_d(const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P, float ds,
                                 uint32_t thresh, uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan, int grid,
                                 const EncHead& H, hipStream_t s) {
    using C = EC<D>;
    const EncTape T = enc_tape_layout(B, S, D, L);
    const size_t ldsb = (size_t)(5 * C::BUF + C::PBUF + 2 * C::PRE) * sizeof(float);
    auto k = enc_fwd_loss_k<D>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
    hipLaunchKernelGGL(k, dim3(grid), dim3(C::NT), ldsb, s, em, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)tape, T, plan, H, seed_dev);
    return re_launch_status();
}

// Training forward + criterion: re_sasrec_encoder_fwd (x0 == NULL, tape given) followed by re_sasrec_loss_rows, as ONE launch.
extern "C" int re_sasrec_encoder_fwd_loss(const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, const int64_t* pos,
                                          const int64_t* neg, int64_t B, int64_t S, int64_t D, int64_t L, const float* const* block_params,
                                          const float* last_w, const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                                          const void* plan, int32_t ncu, float* u, void* tape, size_t tape_bytes, int64_t e_off, int kind,
                                          const int32_t* count, float* loss, float* dU_rows, float* g_rows, int32_t* keys, void* ws,
                                          size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!seq || !pos || !neg || !u || !plan || !tape || !E || !Ptab || !count || !loss || !dU_rows || !g_rows || !keys || !ws || B < 0 || R <= 0)
        return RE_EINVAL;
    if (kind != RE_LOSS_BCE && kind != RE_LOSS_BPR) return RE_EINVAL;
    if ((reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(Ptab) | reinterpret_cast<uintptr_t>(dU_rows) | reinterpret_cast<uintptr_t>(g_rows)) & 15u)
        return RE_EUNSUPPORTED;
    if ((D != 64 && D != 128) || S < 1 || S > 64 || L > SE_MAX_BLOCKS || R >= 0x7FFFFFFFll) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    if (ws_bytes < 256) return RE_EWORKSPACE;
    SasrecParams P;
    if (!se_fill_params(P, block_params, L, last_w, last_b)) return RE_EINVAL;
    if (tape_bytes < (size_t)enc_tape_layout(B, S, D, L).total * sizeof(float)) return RE_EWORKSPACE;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const SeEmbed em{E, Ptab, R, scale};
    const EncHead H{E, R, e_off, pos, neg, kind, count, loss, dU_rows, g_rows, keys, (unsigned long long*)ws};
    if (ncu < 1) ncu = 256;
    const int64_t mt = enc_plan_max_tiles(B, S);
    const int grid = (int)(mt < ncu ? mt : ncu);
    if (D == 128) return enc_fwd_loss_launch_d<128>(em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, grid, H, (hipStream_t)stream);
    return enc_fwd_loss_launch_d<64>(em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, grid, H, (hipStream_t)stream);
}
