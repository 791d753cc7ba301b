// Batch preparation of a SASRec training / evaluation step as ONE device launch (no host sync, capturable):
//   * what the reference does at the top of `fit` (SASRec/main.py:199-204): the mask of non-pad positions, their number M (the
//     denominator of the mean loss), and the destination rows of the 3*B*S item-gradient contributions
//     (seq | pos + 1 | neg + 1, 0 = padding row = dropped) for the scatter-add;
//   * the encoder kernels' work plan (enc_common.h): which (sequence, position) every compact row holds, which tiles form
//     a work item;
//   * optionally copies (seq, pos, neg) into the static buffers a captured step reads and writes the step scalars
//     { seed, 0, lr / (1 - b1^t), 1 / sqrt(1 - b2^t) }.
// Workgroup 0 builds the plan; the other workgroups do the element-wise part.
#define PL_PLAN_KERNEL
#ifdef ENC_PROFILE
__device__ unsigned long long g_plan_marks[4];   // shader-clock stamps of the plan workgroup (100 MHz): spans, ranks, row map
#endif
#include "enc_plan_body.h"

size_t enc_wgrad_part_floats(int64_t D, int64_t L);
size_t enc_wgrad_ppart_floats(int64_t B, int64_t D);
extern "C" size_t re_sasrec_encoder_bwd_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t L);

__global__ __launch_bounds__(PL_NT) void sasrec_batch_prep_k(const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                                             const int64_t* __restrict__ neg, int B, int S, int ncu, int max_tiles, int split_long,
                                                             int64_t* __restrict__ seq_out, int64_t* __restrict__ pos_out,
                                                             int64_t* __restrict__ neg_out, uint8_t* __restrict__ valid,
                                                             int* __restrict__ count, int64_t* __restrict__ rows_all, int* __restrict__ plan,
                                                             uint32_t* __restrict__ state, uint32_t seed, float step_size, float inv_sqrt_bc2,
                                                             PlWeights WP, PlSample SP, PlLoss LA) {
    const int tid = threadIdx.x;
    if (blockIdx.x >= gridDim.x - WP.nblocks) {
        // ---- the tile kernel's weight fragments for this step (re_sasrec_batch_prep_w): the last workgroups of the grid
        const int t = (int)(blockIdx.x - (gridDim.x - WP.nblocks)) * PL_NT + tid;
        if (t < TLC_PREP_THREADS(WP.L, WP.ns)) {
            if (WP.ns == 4) tl_prep_thread<4>(WP.P, WP.L, WP.wf, WP.epoch, t);
            else tl_prep_thread<8>(WP.P, WP.L, WP.wf, WP.epoch, t);
        }
        return;
    }
    if (blockIdx.x > 0) {
        pl_elementwise((int)blockIdx.x - 1, (int)(gridDim.x - 1 - WP.nblocks), seq, pos, neg, B, S, seq_out, pos_out, neg_out, valid, rows_all, SP);
        return;
    }
    // ---- workgroup 0: the step scalars, the previous step's loss into the epoch sum, the plan
    if (threadIdx.x == 0 && LA.acc) LA.acc[0] += LA.prev[0] * LA.w;
    if (threadIdx.x == 0 && state) {
        state[0] = seed;
        state[1] = 0u;
        state[2] = __float_as_uint(step_size);
        state[3] = __float_as_uint(inv_sqrt_bc2);
    }
    __shared__ __align__(16) unsigned char pl_lds[PL_LDS_BYTES];
    pl_plan(seq, B, S, ncu, max_tiles, split_long, count, plan, SP, pl_lds);
}
#ifdef ENC_PROFILE
extern "C" void re_dbg_plan_marks(unsigned long long* out4) { (void)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_plan_marks), 32); }
#endif

extern "C" size_t re_sasrec_plan_bytes(int64_t B, int64_t S) {
    if (B <= 0 || S <= 0) return 256;
    return re_align(enc_plan_bytes(B, S));
}

static int batch_prep_launch(const int64_t* seq, const int64_t* pos, const int64_t* neg, int64_t B, int64_t S, int32_t ncu,
                             int32_t max_tiles, int32_t split_long, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out, uint8_t* valid,
                             int32_t* count, int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed,
                             int64_t step, double lr, double beta1, double beta2, const PlWeights& WP, const PlSample& SP, const PlLoss& LA,
                             re_stream_t stream) {
    re_clear_error();
    if ((!seq && !SP.ptr) || !plan || B <= 0 || S <= 0) return RE_EINVAL;
    if ((pos == nullptr) != (neg == nullptr) || (pos_out == nullptr) != (neg_out == nullptr)) return RE_EINVAL;
    if (S > 64 || B * S > (int64_t)1 << 30 || max_tiles < 1 || max_tiles > 4) return RE_EUNSUPPORTED;
    if (plan_bytes < enc_plan_bytes(B, S)) return RE_EWORKSPACE;
    if (state && step < 1) return RE_EINVAL;
    if (ncu < 1) ncu = 1;
    float ss = 0.f, ib = 0.f;
    if (state) {
        ss = (float)(lr / (1.0 - pow(beta1, (double)step)));
        ib = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
    }
    const bool elementwise = seq_out || valid || rows_all || pos_out;
    const unsigned grid = 1 + (elementwise ? re_grid(B * S, PL_NT, 256) : 0) + (unsigned)WP.nblocks;
    hipLaunchKernelGGL(sasrec_batch_prep_k, dim3(grid), dim3(PL_NT), 0, (hipStream_t)stream, seq, pos, neg, (int)B, (int)S, (int)ncu,
                       (int)max_tiles, (int)split_long, seq_out, pos_out, neg_out, valid, count, rows_all, (int*)plan, state, seed, ss, ib, WP, SP, LA);
    return re_launch_status();
}

extern "C" int re_sasrec_batch_prep(const int64_t* seq, const int64_t* pos, const int64_t* neg, int64_t B, int64_t S, int32_t ncu,
                                    int32_t max_tiles, int32_t split_long, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out, uint8_t* valid,
                                    int32_t* count, int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed,
                                    int64_t step, double lr, double beta1, double beta2, re_stream_t stream) {
    PlWeights WP{};
    WP.nblocks = 0;
    PlSample SP{};
    PlLoss LA{};
    return batch_prep_launch(seq, pos, neg, B, S, ncu, max_tiles, split_long, seq_out, pos_out, neg_out, valid, count, rows_all, plan, plan_bytes, state,
                             seed, step, lr, beta1, beta2, WP, SP, LA, stream);
}

// The same launch + the weight preparation of the D = 64 one-tile-per-workgroup step (re_sasrec_encoder_step_part, part + 8) in extra
// workgroups: the encoder's matrices as bf16 hi / mid fragment planes into the backward workspace `ws`, the launch epoch in `tape`'s flag
// area advanced -- the step that follows on the same stream then needs no preparation launch of its own.
static int pl_fill_weights(PlWeights& WP, const float* const* block_params, const float* last_w, const float* last_b, int64_t L, int64_t D, int64_t B,
                           int64_t S, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes) {
    WP.nblocks = 0;
    if (!block_params) return RE_OK;
    if (!enc_tile_width_ok(D) || !tape || !ws || B <= 0 || S <= 0 || S > 64) return RE_EUNSUPPORTED;
    if (!se_fill_params(WP.P, block_params, L, last_w, last_b)) return RE_EINVAL;
    if (tape_bytes < (size_t)enc_tape_layout(B, S, D, L).total * sizeof(float) || ws_bytes < re_sasrec_encoder_bwd_workspace_bytes(B, S, D, L))
        return RE_EWORKSPACE;
    // (the workspace layout of re_sasrec_encoder_step: slab | weight-gradient partials | gradient tape | fragments | inboxes)
    WP.L = (int)L;
    WP.ns = (int)(D / 16);
    WP.nblocks = (TLC_PREP_THREADS((int)L, WP.ns) + PL_NT - 1) / PL_NT;
    WP.wf = enc_bwd_ws(ws, B, S, D, L).wf;
    WP.epoch = enc_tile_epoch(tape, B, S, L, D);
    return RE_OK;
}

extern "C" int re_sasrec_batch_prep_w(const int64_t* seq, const int64_t* pos, const int64_t* neg, int64_t B, int64_t S, int32_t ncu,
                                      int32_t max_tiles, int32_t split_long, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out, uint8_t* valid,
                                      int32_t* count, int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed,
                                      int64_t step, double lr, double beta1, double beta2, const float* const* block_params, const float* last_w,
                                      const float* last_b, int64_t L, int64_t D, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                                      const float* prev_loss, float* loss_acc, float loss_weight, re_stream_t stream) {
    PlWeights WP{};
    const int rc = pl_fill_weights(WP, block_params, last_w, last_b, L, D, B, S, tape, tape_bytes, ws, ws_bytes);
    if (rc != RE_OK) return rc;
    if ((prev_loss == nullptr) != (loss_acc == nullptr)) return RE_EINVAL;
    PlSample SP{};
    PlLoss LA{prev_loss, loss_acc, loss_weight};
    return batch_prep_launch(seq, pos, neg, B, S, ncu, max_tiles, split_long, seq_out, pos_out, neg_out, valid, count, rows_all, plan, plan_bytes, state,
                             seed, step, lr, beta1, beta2, WP, SP, LA, stream);
}

// SAMPLE + PREPARE as one launch: the batch is row b = user order[b0 + b] of the SASRec training chain (re_seq_train_sample: the same
// rows and draws, sample_seed / sample_step as there) written straight into the static buffers a captured step reads (seq_out / pos_out /
// neg_out must be given), with everything re_sasrec_batch_prep derives from it; block_params != NULL: + the tile step's weight fragments
// (re_sasrec_batch_prep_w).  users (optional) receives the rows' user ids.
extern "C" int re_seq_train_sample_prep(const int64_t* ptr, const int64_t* items, const int64_t* sorted_items, const int64_t* order, int64_t n_order,
                                        int64_t b0, int64_t N, uint32_t sample_seed, uint32_t sample_step, int64_t* users, int64_t B, int64_t S,
                                        int32_t ncu, int32_t max_tiles, int32_t split_long, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out,
                                        uint8_t* valid, int32_t* count, int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed,
                                        int64_t step, double lr, double beta1, double beta2, const float* const* block_params, const float* last_w,
                                        const float* last_b, int64_t L, int64_t D, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                                        const float* prev_loss, float* loss_acc, float loss_weight, re_stream_t stream) {
    if (!ptr || !items || !sorted_items || !order || !seq_out || !pos_out || !neg_out || n_order < 0 || b0 < 0 || N < 1) return RE_EINVAL;
    if (B <= 0 || S <= 0 || S > 64 || B * S > (int64_t)1 << 26) return RE_EUNSUPPORTED;            // (32-bit draw counter: position * 32)
    PlSample SP{ptr, items, sorted_items, order, n_order, b0, N, sample_seed, sample_step, users};
    PlWeights WP{};
    const int rc = pl_fill_weights(WP, block_params, last_w, last_b, L, D, B, S, tape, tape_bytes, ws, ws_bytes);
    if (rc != RE_OK) return rc;
    if ((prev_loss == nullptr) != (loss_acc == nullptr)) return RE_EINVAL;
    PlLoss LA{prev_loss, loss_acc, loss_weight};
    return batch_prep_launch(nullptr, pos_out, neg_out, B, S, ncu, max_tiles, split_long, seq_out, pos_out, neg_out, valid, count, rows_all, plan,
                             plan_bytes, state, seed, step, lr, beta1, beta2, WP, SP, LA, stream);
}

// ---- the launch in FRONT of a captured step whose batch was prepared by the previous step's tail launch (re_next_prep, enc_tail.hip): what is
//      left of the preparation launch -- the step scalars, the previous loss into the epoch sum, the tile kernels' weight fragments (they need
//      the parameters the previous step's optimizer left) -- and the mailbox: where the FOLLOWING batch lives (seq = NULL: there is none).
__global__ __launch_bounds__(PL_NT) void sasrec_step_stage_k(uint32_t* __restrict__ state, uint32_t seed, float step_size, float inv_sqrt_bc2,
                                                             PlWeights WP, PlLoss LA, PlMail* __restrict__ mail, PlMail M) {
    const int tid = threadIdx.x;
    if (blockIdx.x > 0) {
        const int t = ((int)blockIdx.x - 1) * PL_NT + tid;
        if (t < TLC_PREP_THREADS(WP.L, WP.ns)) {
            if (WP.ns == 4) tl_prep_thread<4>(WP.P, WP.L, WP.wf, WP.epoch, t);
            else tl_prep_thread<8>(WP.P, WP.L, WP.wf, WP.epoch, t);
        }
        return;
    }
    if (tid != 0) return;
    if (LA.acc) LA.acc[0] += LA.prev[0] * LA.w;
    if (state) {
        state[0] = seed;
        state[1] = 0u;
        state[2] = __float_as_uint(step_size);
        state[3] = __float_as_uint(inv_sqrt_bc2);
    }
    if (mail) *mail = M;
}

static int step_stage_launch(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, void* mail, const PlMail& M, int64_t B,
                             int64_t S, const float* const* block_params, const float* last_w, const float* last_b, int64_t L, int64_t D, void* tape,
                             size_t tape_bytes, void* ws, size_t ws_bytes, const float* prev_loss, float* loss_acc, float loss_weight,
                             re_stream_t stream) {
    re_clear_error();
    if (!state || step < 1) return RE_EINVAL;
    if ((M.seq || M.SP.ptr) && !mail) return RE_EINVAL;
    PlMail Me = M;
    Me.epoch = (unsigned)(step & 0x3FFFFFFF) + 2u;          // (the span hand-over flag's value for this step: never 0, 1 or the value of two steps ago)
    if ((prev_loss == nullptr) != (loss_acc == nullptr)) return RE_EINVAL;
    PlWeights WP{};
    const int rc = pl_fill_weights(WP, block_params, last_w, last_b, L, D, B, S, tape, tape_bytes, ws, ws_bytes);
    if (rc != RE_OK) return rc;
    const PlLoss LA{prev_loss, loss_acc, loss_weight};
    const float ss = (float)(lr / (1.0 - pow(beta1, (double)step))), ib = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
    hipLaunchKernelGGL(sasrec_step_stage_k, dim3(1 + (unsigned)WP.nblocks), dim3(PL_NT), 0, (hipStream_t)stream, state, seed, ss, ib, WP, LA,
                       (PlMail*)mail, Me);
    return re_launch_status();
}

extern "C" int re_sasrec_step_stage(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, void* mail,
                                    const int64_t* next_seq, const int64_t* next_pos, const int64_t* next_neg, int64_t B, int64_t S,
                                    const float* const* block_params, const float* last_w, const float* last_b, int64_t L, int64_t D, void* tape,
                                    size_t tape_bytes, void* ws, size_t ws_bytes, const float* prev_loss, float* loss_acc, float loss_weight,
                                    re_stream_t stream) {
    if (next_seq && (!next_pos || !next_neg)) return RE_EINVAL;
    PlMail M{};
    M.seq = next_seq; M.pos = next_pos; M.neg = next_neg;
    return step_stage_launch(state, seed, step, lr, beta1, beta2, mail, M, B, S, block_params, last_w, last_b, L, D, tape, tape_bytes, ws, ws_bytes,
                             prev_loss, loss_acc, loss_weight, stream);
}

// The same with a SAMPLING source for the next batch (re_seq_train_sample_prep's: rows b0 .. b0 + B of `order`, sample_seed / sample_step);
// ptr = NULL: no next batch.
extern "C" int re_sasrec_step_stage_sample(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, void* mail,
                                           const int64_t* ptr, const int64_t* items, const int64_t* sorted_items, const int64_t* order,
                                           int64_t n_order, int64_t b0, int64_t N, uint32_t sample_seed, uint32_t sample_step, int64_t* users,
                                           int64_t B, int64_t S, const float* const* block_params, const float* last_w, const float* last_b,
                                           int64_t L, int64_t D, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, const float* prev_loss,
                                           float* loss_acc, float loss_weight, re_stream_t stream) {
    PlMail M{};
    if (ptr) {
        if (!items || !sorted_items || !order || n_order < 0 || b0 < 0 || N < 1) return RE_EINVAL;
        if (B <= 0 || S <= 0 || S > 64 || B * S > (int64_t)1 << 26) return RE_EUNSUPPORTED;
        M.SP = PlSample{ptr, items, sorted_items, order, n_order, b0, N, sample_seed, sample_step, users};
    }
    return step_stage_launch(state, seed, step, lr, beta1, beta2, mail, M, B, S, block_params, last_w, last_b, L, D, tape, tape_bytes, ws, ws_bytes,
                             prev_loss, loss_acc, loss_weight, stream);
}
