/* recengine -- C ABI of the MI355X-native embedding-and-scoring engine (librecengine.so).
 *
 * This is the drop-in boundary for the hot path named in BASELINE.json `north_star`: every entry point
 * replaces one group of aten ops that the reference's model scripts dispatch through PyTorch
 * (the reference has no native code and no FFI of its own -- SURVEY.md §2b -- so "the reference interface
 * each one replaces" is the Python call site cited per function, paths relative to MTandHJ/RecBoard).
 *
 * Conventions (SURVEY.md §8b "C ABI underneath"):
 *   - plain pointers + sizes; all pointers are DEVICE pointers unless named `h_*`; no torch types.
 *   - the CALLER owns every buffer; kernels allocate nothing.  Ops that need scratch take `ws`/`ws_bytes`
 *     and have a `*_workspace_bytes()` query.
 *   - stream-ordered on `stream` (a hipStream_t passed as void*); no internal device synchronisation;
 *     no global mutable state (re-entrant).
 *   - return 0 on success, a negative RE_E* code otherwise (no exceptions cross the ABI).
 *   - fp32 data, int64 indices (the reference is fp32/int64 everywhere).
 *   - out-of-range indices never fault: they read as a zero row / are dropped.
 */
#ifndef RECENGINE_H
#define RECENGINE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RE_OK 0
#define RE_EINVAL (-1)    /* bad argument (null pointer, negative size, unsupported D/K) */
#define RE_EWORKSPACE (-2) /* workspace too small */
#define RE_ELAUNCH (-3)   /* hipGetLastError() != hipSuccess after a launch */
#define RE_EUNSUPPORTED (-4)

#define RE_TOPK_MAX 64          /* K <= 64 (cfg.monitors use K <= 50) */
#define RE_MASKED_SCORE (-1e23f) /* UniSRec/main.py:413 `scores[seen] = -1e23` */

typedef void* re_stream_t; /* hipStream_t */

int re_abi_version(void);
const char* re_error_string(int code);
/* resident workgroups per CU the one-tile-per-workgroup step kernels of this build hold (1): what `split_long & 8` of re_sasrec_batch_prep
 * must agree with (the plan's residency rule counts them). */
int re_tile_wgs_per_cu(int64_t D);
/* 1: every plan of this shape made with split_long = 4 (| 8) -- no splitting, tile kernels forced -- gives the step to the tile kernels
 * (plan header word 7), so re_sasrec_encoder_step_part may be called without its workgroup-per-item launch (part & 2 clear). */
int re_sasrec_tile_step_certain(int64_t B, int64_t S, int64_t D);

/* ---------------------------------------------------------------------------------------------------------
 * K1  embedding row gather.  out[i, :] = W[idx[i], :]   (i < n)
 * Replaces `nn.Embedding.__call__` / `W[idx]`:  SASRec/main.py:183,200-204; MF-BPR/main.py:84-86;
 * LightGCN/main.py:91-93,101-103; DeepFM/main.py:59-61,204-206.
 * HBM-bound: 8 + 8*D algorithmic bytes per looked-up row. */
int re_gather_rows(const float* W, int64_t R, int64_t D, const int64_t* idx, int64_t n, float* out,
                   re_stream_t stream);

/* SASRec front end, fused: out[b,s,:] = seq[b,s]==0 ? 0 : dropout(E[seq[b,s]] * scale + P[s])
 * Replaces SASRec/main.py:181-187 (embedding, `*= D**0.5`, mark_position, embdDropout, masked_fill).
 * drop_p == 0 disables dropout; otherwise the engine's counter-based mask (stream id 1, element id
 * (b*S+s)*D+d, see csrc/re_rng.h) scaled by 1/(1-p).  Every dropout entry point takes the seed twice: `seed` by value
 * and `seed_dev` (DEVICE uint32[1], may be NULL); the effective seed is seed ^ *seed_dev, so a step captured in a hipGraph
 * can be replayed with a fresh per-step seed uploaded to device memory. */
int re_sasrec_embed(const float* E, int64_t R, int64_t D, const float* P, const int64_t* seq, int64_t B,
                    int64_t S, float scale, float drop_p, uint32_t seed, const uint32_t* seed_dev, float* out,
                    re_stream_t stream);

/* Backward of re_sasrec_embed, in place on gx [B,S,D]: in = gradient w.r.t. x0 (from re_sasrec_encoder_bwd), out =
 * contribution rows for re_scatter_add_rows (pad rows zero, the forward's dropout mask re-applied, times `scale`);
 * dP [S,D] = sum over b of the masked gradient (gradient of the position table, SASRec/main.py:159-161).
 * Deterministic. */
size_t re_sasrec_embed_bwd_workspace_bytes(int64_t S, int64_t D);
int re_sasrec_embed_bwd(float* gx, const int64_t* seq, int64_t B, int64_t S, int64_t D, float scale, float drop_p,
                        uint32_t seed, const uint32_t* seed_dev, float* dP, void* ws, size_t ws_bytes, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K1b  dense gradient of the gather: dW[r,:] = sum_{i: idx[i]==r} g[i,:], rows == padding_idx skipped,
 * every other row zero.  dW [R,D] is fully overwritten (accumulate = 0) or added to (accumulate = 1).
 * Deterministic (sorted segments, fixed chunking):
 * two calls on the same input give bit-identical output.
 * Replaces aten embedding_dense_backward / index_put_(accumulate) reached from `loss.backward()`
 * (SASRec/main.py:249, MF-BPR/main.py:122).  `scale` multiplies every contribution (SASRec: sqrt(D)). */
/* The two halves of re_scatter_add_rows as separate entry points (same workspace, same sizes):
 *   re_scatter_plan   -- the index half: stable sort of (destination row, position) into ws; depends on idx only, so a
 *                        training step can run it on a second stream while the gradient rows are still being produced.
 *                        Optionally zero-fills `zero_fill[0:zero_floats]` (the table the apply step accumulates into).
 *   re_scatter_apply  -- the data half: segmented sum of the rows of g in sorted order into dW (accumulate = 1: dW += ...; 0: dW is
 *                        zero-filled first; 2: only the rows that occur are written -- assigned -- and the others keep their contents).
 *                        Must follow a re_scatter_plan with the same (n, D, R, ws).
 * plan followed by apply(accumulate = 1) on a zero-filled table equals re_scatter_add_rows(accumulate = 0). */
int re_scatter_plan(const int64_t* idx, int64_t n, int64_t D, int64_t R, int64_t padding_idx, float* zero_fill,
                    int64_t zero_floats, void* ws, size_t ws_bytes, re_stream_t stream);
int re_scatter_apply(const float* g, int64_t n, int64_t D, int64_t R, float scale, float* dW, int accumulate, void* ws,
                     size_t ws_bytes, re_stream_t stream);
/* Row-sparse Adam for tables whose dense gradient does not fit: for every distinct destination row r
 * of idx (padding_idx / out-of-range entries dropped) G_r = sum of the rows of g pointing at r (position order, deterministic),
 * then ONE Adam update of row r of (W, m, v) -- torch.optim.SparseAdam's rule (global step count in the bias corrections) plus
 * coupled weight decay g += wd * w on the touched rows.  Rows without a gradient are not touched.  Workspace as
 * re_scatter_add_rows (same n, D, R). */
int re_sparse_adam_rows(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R, int64_t padding_idx, float* W,
                        float* m, float* v, int64_t step, double lr, double beta1, double beta2, double eps,
                        double weight_decay, void* ws, size_t ws_bytes, re_stream_t stream);
/* hipGraph-friendly form of re_sparse_adam_rows: hyper (DEVICE float[2]) = { lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) } as for
 * re_adam_step_dev. */
int re_sparse_adam_rows_dev(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R, int64_t padding_idx, float* W,
                            float* m, float* v, const float* hyper, double beta1, double beta2, double eps,
                            double weight_decay, void* ws, size_t ws_bytes, re_stream_t stream);
/* The same update for a SMALL key list (a training step's contribution rows: ~15 k at config 5) against a table of any size, in ONE
 * launch and without a sort or a workspace (csrc/adam_rows.hip: the workgroup a key hashes to collects, orders and sums that key's rows).
 *   keys: int32 (key_bytes = 4) or int64 (8), n_regions regions of region_stride entries each; of every region the first
 *   min(n_dev[0] * n_mul, region_stride) entries are read when n_dev (DEVICE int32) is given, else the first n_host;
 *   g: one row per key entry, [n_regions * region_stride, D]; entries < 0, >= R or == padding_idx are dropped; D = 64 or 128.
 *   hyper (DEVICE float[2], as re_sparse_adam_rows_dev) or, when null, the host's step / lr.
 * The sums' association differs from re_sparse_adam_rows' (same values to rounding); deterministic for a given input. */
int re_sparse_adam_rows_small(const float* g, const void* keys, int32_t key_bytes, int32_t n_regions, int64_t region_stride,
                              const int32_t* n_dev, int64_t n_mul, int64_t n_host, int64_t D, int64_t R, int64_t padding_idx, float* W,
                              float* m, float* v, const float* hyper, int64_t step, double lr, double beta1, double beta2, double eps,
                              double weight_decay, re_stream_t stream);
/* Optional: the dense Adam of the parameters whose gradients a launch FINISHES, applied by that launch (no optimizer launch of its
 * own behind it).  grad_base / param / m / v are arenas of one layout: the gradient written at grad_base + i updates element i.
 * hyper = device { lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) } (re_step_state / re_sasrec_batch_prep); {0, 0} = leave everything as it is. */
typedef struct re_adam_fuse {
    const float* grad_base;
    float *param, *m, *v;
    const float* hyper;
    double beta1, beta2, eps, weight_decay;   /* (doubles, as re_adam_step takes them: 1 - beta is formed in double) */
} re_adam_fuse;

/* Small dense tables (R up to ~100 k rows; D = 64 or 128): the same sum WITHOUT the sort -- every workgroup owns a range of
 * destination rows, scans all keys and adds the rows that fall into its range; one launch, no workspace, dW [R, D] fully
 * overwritten (untouched rows and row `padding_idx` zero).  keys are int32: `n_regions` runs of n keys, run q at
 * keys[q * region_stride], its rows at g[(q * region_stride + i) * D]; n = n_dev[0] * n_mul read on the device (a captured
 * launch follows the batch: e.g. the tile count of re_sasrec_batch_prep's plan, * 16) or n_host when n_dev is NULL.
 * Deterministic (fixed summation tree; not the tree of re_scatter_add_rows).  Replaces the same aten
 * embedding_dense_backward as re_scatter_add_rows, for the three contribution sets of a SASRec step at once. */
int re_scatter_add_rows_small(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev,
                              int32_t n_mul, int64_t n_host, int64_t D, int64_t R, int64_t padding_idx, float scale, float* dW,
                              re_stream_t stream);
/* The same sum with the table's DENSE Adam update folded in: every row's owner has the row's finished gradient in hand, so it
 * updates adam->param / m / v rows [0, R) right there (every row moves, as torch.optim.Adam's dense step does) -- the gradient
 * is not written (dW may be NULL) and no optimizer launch over the table follows.  adam->grad_base is unused here. */
int re_scatter_adam_rows_small(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev,
                               int32_t n_mul, int64_t n_host, int64_t D, int64_t R, int64_t padding_idx, float scale, float* dW,
                               const re_adam_fuse* adam, re_stream_t stream);
/* The NEXT batch's preparation riding in a step's tail launch (re_sasrec_step_tail / _sparse): what re_sasrec_batch_prep computes -- mask,
 * count, destination rows, the encoder's plan, the staged copies -- for the batch whose addresses `mail` holds, into the static buffers of the
 * captured step that will consume it.  mail: RE_MAIL_BYTES of device memory, written by re_sasrec_step_stage (the batch's tensors) or
 * re_sasrec_step_stage_sample (its sampling source: the batch is drawn as re_seq_train_sample_prep draws it) in front of the step; neither
 * given: no next batch, nothing is prepared.  The outputs are re_sasrec_batch_prep's (a sampled batch needs seq_out / pos_out / neg_out). */
#define RE_MAIL_BYTES 128
typedef struct {
    const void* mail;
    int64_t B, S;
    int32_t ncu, max_tiles, split_long;
    void *seq_out, *pos_out, *neg_out, *valid, *count, *rows_all, *plan;
    size_t plan_bytes;
} re_next_prep;
/* The launch in front of a captured step whose batch a previous tail launch prepared: step scalars into `state` (as re_sasrec_batch_prep),
 * acc += prev_loss * weight (optional), the tile kernels' weight fragments (block_params != NULL: as re_sasrec_batch_prep_w), and the mailbox
 * for THIS step's tail launch (next_seq = NULL: no next batch). */
int re_sasrec_step_stage(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, void* mail, const int64_t* next_seq,
                         const int64_t* next_pos, const int64_t* next_neg, int64_t B, int64_t S, const float* const* block_params,
                         const float* last_w, const float* last_b, int64_t L, int64_t D, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                         const float* prev_loss, float* loss_acc, float loss_weight, re_stream_t stream);
int re_sasrec_step_stage_sample(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, void* mail, const int64_t* ptr,
                                const int64_t* items, const int64_t* sorted_items, const int64_t* order, int64_t n_order, int64_t b0, int64_t N,
                                uint32_t sample_seed, uint32_t sample_step, int64_t* users, int64_t B, int64_t S, const float* const* block_params,
                                const float* last_w, const float* last_b, int64_t L, int64_t D, void* tape, size_t tape_bytes, void* ws,
                                size_t ws_bytes, const float* prev_loss, float* loss_acc, float loss_weight, re_stream_t stream);
/* The tail of a D = 64 SASRec training step as one launch + the reduction: re_scatter_adam_rows_small (or, table_adam NULL,
 * re_scatter_add_rows_small; scale 1, n from n_dev) over the step's contribution rows, whose 1024-thread workgroups then take the jobs of
 * re_sasrec_encoder_step_part(part = 4) -- the weight gradients of the encoder from the tape the item kernels left in `tape` / `ws` -- from
 * a ticket counter; enc_adam (optional) as there.  Both halves depend on the item kernels alone: one queue, no fork and join around them
 * (csrc/enc_tail.hip).  Results bit-identical to the two calls.  `ticket`: one zero-initialised uint32 of the caller's, left zero.
 * next (optional, re_next_prep): the launch also prepares the next batch.
 * Replaces: embedding_dense_backward of the item table + the autograd weight gradients of SASRec/main.py:170-197's blocks + the optimizer
 * step over both (SASRec/main.py:249-252). */
int re_sasrec_step_tail(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev, int32_t n_mul,
                        int64_t R, int64_t padding_idx, float* dW, const re_adam_fuse* table_adam, const int64_t* seq, int64_t B, int64_t S,
                        int64_t D, int64_t L, const void* plan, int32_t ncu, const void* tape, size_t tape_bytes, const float* dx0,
                        float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b, void* ws, size_t ws_bytes,
                        const re_adam_fuse* enc_adam, uint32_t* ticket, const re_next_prep* next, re_stream_t stream);
/* The same tail for a LARGE item table (config 5): re_sparse_adam_rows_small (int32 keys, hyper from device memory) whose workgroups then take
 * the weight-gradient jobs; D = 64 or 128. */
int re_sasrec_step_tail_sparse(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev, int64_t n_mul,
                               int64_t R, int64_t padding_idx, float* W, float* m, float* v, const float* hyper, double beta1, double beta2,
                               double eps, double weight_decay, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L, const void* plan,
                               int32_t ncu, const void* tape, size_t tape_bytes, const float* dx0, float emb_scale, float* dPtab,
                               float* const* block_grads, float* g_last_w, float* g_last_b, void* ws, size_t ws_bytes,
                               const re_adam_fuse* enc_adam, uint32_t* ticket, const re_next_prep* next, re_stream_t stream);
size_t re_scatter_add_rows_workspace_bytes(int64_t n, int64_t D, int64_t R);
int re_scatter_add_rows(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R,
                        int64_t padding_idx, float scale, float* dW, int accumulate, void* ws, size_t ws_bytes,
                        re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K3  fused pair-logit criteria over table rows.
 *
 * re_pair_loss_fwd: for every position i < n with valid[i] != 0 (valid may be NULL = all):
 *      pl = <U[i,:], E[e_off + pos[i], :]>,  nl = <U[i,:], E[e_off + neg[i], :]>
 *      kind RE_LOSS_BCE: l_i = softplus(-pl) + softplus(nl)     (BCE(pl,1) + BCE(nl,0), SASRec/main.py:208-214)
 *      kind RE_LOSS_BPR: l_i = softplus(nl - pl)                (SASRec/main.py:215; MF-BPR/main.py:88-91)
 *   loss[0] = sum_i l_i / M,  M = #valid positions (reduction="mean"); pl/nl are kept in `logits[2n]`
 *   for the backward.  `count` (int32[1]) receives M.
 * re_pair_loss_bwd: given dloss (device scalar, may be NULL = 1.0) writes
 *      dU[i,:] = dpl*E[pos] + dnl*E[neg]   (zero row where !valid)
 *      gpos[i,:] = dpl*U[i,:], gneg[i,:] = dnl*U[i,:]   (contribution rows for re_scatter_add_rows)
 * U is [n, D] with row stride ldu (floats) so that userEmbds [B,S,D] can be passed un-compacted:
 * the boolean-mask compaction of SASRec/main.py:199-204 (a host sync in the reference) is fused away. */
#define RE_LOSS_BCE 0
#define RE_LOSS_BPR 1
size_t re_pair_loss_workspace_bytes(int64_t n);
int re_pair_loss_fwd(const float* U, int64_t ldu, const float* E, int64_t R, int64_t D, int64_t e_off,
                     const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                     float* logits, float* loss, int32_t* count, void* ws, size_t ws_bytes,
                     re_stream_t stream);
int re_pair_loss_bwd(const float* U, int64_t ldu, const float* E, int64_t R, int64_t D, int64_t e_off,
                     const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                     const float* logits, const int32_t* count, const float* dloss, float* dU, int64_t lddu,
                     float* gpos, float* gneg, re_stream_t stream);

/* Forward + backward in one pass for training steps (upstream gradient of the mean loss = 1): `count` (DEVICE int32[1]) must
 * hold M, the number of valid positions -- known at batch assembly (sum of `valid`), so no reduction has to finish before
 * the gradient rows are scaled.  Outputs as re_pair_loss_fwd (loss) and re_pair_loss_bwd (dU, gpos, gneg). */
int re_pair_loss_fwd_bwd(const float* U, int64_t ldu, const float* E, int64_t R, int64_t D, int64_t e_off,
                         const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                         const int32_t* count, float* loss, float* dU, int64_t lddu, float* gpos, float* gneg,
                         void* ws, size_t ws_bytes, re_stream_t stream);

/* MF-BPR / LightGCN triplet form (MF-BPR/main.py:81-93): rows gathered from TWO tables inside the kernel.
 *      pl = <Ut[users[i]], It[pos[i]]>, nl = <Ut[users[i]], It[neg[i]]>, loss = mean softplus(nl - pl)
 * bwd writes the three contribution-row sets (for re_scatter_add_rows into dUt / dIt). */
int re_bpr_triplet_fwd(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D,
                       const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                       float* logits, float* loss, void* ws, size_t ws_bytes, re_stream_t stream);
int re_bpr_triplet_bwd(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D,
                       const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                       const float* logits, const float* dloss, float* gu, float* gpos, float* gneg,
                       re_stream_t stream);
/* forward + backward in one pass (M = n) */
int re_bpr_triplet_fwd_bwd(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D, const int64_t* users,
                           const int64_t* pos, const int64_t* neg, int64_t n, float* loss, float* gu, float* gpos,
                           float* gneg, void* ws, size_t ws_bytes, re_stream_t stream);
/* The same with its outputs laid out for ONE owner-computes update of the user | item arena (re_scatter_adam_rows_small): g [3][n][D] =
 * the gradient rows of the user / positive / negative lookups, keys int32 [3][n] = their rows in a table of RU user rows followed by RI
 * item rows (-1: no contribution).  Replaces the three `index` backward scatters + Adam of MF-BPR/main.py:116-123 by two launches. */
int re_bpr_triplet_step_rows(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D, const int64_t* users,
                             const int64_t* pos, const int64_t* neg, int64_t n, float* loss, float* g, int32_t* keys, void* ws,
                             size_t ws_bytes, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K4  full-catalog scoring.
 * re_score_dense: out[b,n] = <Q[b,:], E[n,:]>   -- literal drop-in for `recommend_from_full`
 *   (SASRec/main.py:228 einsum("BD,ND->BN"); MF-BPR/main.py:104; LightGCN/main.py:120).
 * re_score_topk: the Coach.evaluate contract (freerec, mirrored at UniSRec/main.py:408-414) fused:
 *   scores -> scores[seen] = -1e23 -> top-K, without materialising B x N.
 *   vals [B,K] sorted descending, idx [B,K] int64; ties -> lowest item index; if fewer than K items are
 *   unmasked the tail is filled with masked items (value -1e23) in ascending index order; if K > N the
 *   remaining slots get (-inf, -1).  seen_ptr[B+1]/seen_idx[nnz] is a CSR of int64 item ids, ASCENDING inside
 *   every user's range (duplicates allowed);
 *   seen_ptr == NULL means retain_seen.
 * Arithmetic: every returned value is the k-ordered fp32 chain acc = fmaf(q[k], e[k], acc) -- exact fp32, and the indices are
 *   the exact top K under (value, lowest index).  For D = 64 / 128 and K <= 50 (many users or a long catalog) the catalog is first screened with bf16 hi/mid
 *   split products on the XDL matrix pipe (v_mfma_f32_32x32x16_bf16, 3 products per 16 k), the K + 6 best candidates of every
 *   user are re-scored exactly and the result is certified against a rigorous error bound; users that cannot be certified
 *   are redone by the exact fp32-MFMA kernel (v_mfma_f32_32x32x2_f32) in the same call.  Same results either way.
 * D must be a multiple of 8 and <= 256; K <= RE_TOPK_MAX.
 * The workspace (and a prep buffer) must be 16-byte aligned.
 * re_score_prepare / re_score_topk_prepared: the item table's split planes are built once (prep buffer of
 *   re_score_prepare_bytes(N, D) bytes, D = 64 or 128) and reused by any number of scoring calls against the same table --
 *   Coach.evaluate scores every user batch of a split against one table (UniSRec/main.py:400-447).  E must be the same
 *   table the planes were made from (it is read for the exact re-scoring). */
int re_score_dense(const float* Q, const float* E, int64_t B, int64_t N, int64_t D, float* out,
                   re_stream_t stream);
size_t re_score_topk_workspace_bytes(int64_t B, int64_t N, int64_t D, int64_t K);
int re_score_topk(const float* Q, const float* E, int64_t B, int64_t N, int64_t D,
                  const int64_t* seen_ptr, const int64_t* seen_idx, int64_t K, float* vals, int64_t* idx,
                  void* ws, size_t ws_bytes, re_stream_t stream);
size_t re_score_prepare_bytes(int64_t N, int64_t D);
int re_score_prepare(const float* E, int64_t N, int64_t D, void* prep, size_t prep_bytes, re_stream_t stream);
size_t re_score_topk_prepared_workspace_bytes(int64_t B, int64_t N, int64_t D, int64_t K);
int re_score_topk_prepared(const float* Q, const float* E, const void* prep, int64_t B, int64_t N, int64_t D,
                           const int64_t* seen_ptr, const int64_t* seen_idx, int64_t K, float* vals, int64_t* idx,
                           void* ws, size_t ws_bytes, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Owner bucketing of a batch's table lookups for a row-sharded table (row r lives on rank r mod G; freerec has no
 * counterpart -- the reference trains one replicated nn.Embedding, MF-BPR/main.py:36-42): one stable counting-sort pass over the G
 * owners, no host sync, FIXED capacity per peer -- equal-split all-to-alls, capturable.
 *   buckets [G * cap] int64: the LOCAL row ids (r div G) wanted from owner g at [g * cap, ...), in order of appearance, -1 in unused slots;
 *   slot [n] int64: g * cap + rank of lookup j inside its bucket (gather the received rows / scatter the gradient rows by it), -1 if dropped;
 *   counts [G + 1] int32: lookups per owner (may exceed cap), then the number of DROPPED lookups (index outside [0, R), or bucket
 *   full) -- the caller checks counts[G] == 0 at its next sync point.
 *   skip_row >= 0: lookups of that row (the padding row: most of a left-padded batch, SASRec/main.py:143-157) take no slot and are not
 *   counted as dropped; their slot is -1.  -1: every in-range lookup takes a slot. */
size_t re_route_workspace_bytes(int64_t n, int64_t G);
int re_route_bucket(const int64_t* idx, int64_t n, int64_t R, int64_t G, int64_t cap, int64_t skip_row, int64_t* buckets, int64_t* slot,
                    int32_t* counts, void* ws, size_t ws_bytes, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Batch preparation of a SASRec step (what the top of `fit` does, SASRec/main.py:199-204, plus the encoder's work plan), as
 * ONE device launch with no host sync (capturable):
 *   valid [B*S] u8 = seq != 0;  count int32[1] = number of valid positions (M of the mean loss);
 *   rows_all int64 [3*B*S] = seq | (valid ? pos + 1 : 0) | (valid ? neg + 1 : 0): destination rows of the step's item-gradient
 *   contributions (0 = padding row = dropped by re_scatter_add_rows);
 *   plan (re_sasrec_plan_bytes(B, S) bytes): the encoder kernels' work items.  Sequences are left-padded (SASRec/main.py:143-157),
 *   so only the rows from a sequence's first real token on are materialised, in tiles of 16 rows: sequences of <= 16 rows share
 *   tiles (power-of-two slots), longer ones own 2..4 tiles; a work item is one long sequence or up to `max_tiles` tiles of short
 *   ones, chosen so that the items just fill `ncu` compute units.  The pad positions in front of a sequence are identical keys
 *   (k = b_k, v = b_v) and enter the softmax analytically (one virtual key with multiplicity), forward and backward.
 *   Optional: seq_out / pos_out / neg_out receive copies of the batch (static buffers of a captured step); state (DEVICE
 *   uint32[4]) = { seed, 0, bits(lr / (1 - beta1^step)), bits(1 / sqrt(1 - beta2^step)) } -- `state` doubles as `seed_dev` of the dropout entry points and, from word 2, as `hyper` of re_adam_step_dev.
 *   pos / neg may be NULL (evaluation: only the plan is wanted); then valid / count / rows_all rows 1, 2 are not written.
 *   split_long is a mask.  & 2: the plan must NOT hand the step to the one-tile-per-workgroup kernels (re_sasrec_encoder_step: hdr[7] stays 0).
 *   & 4: the plan MUST hand it to them where their grid allows.  Otherwise: batches of up to 2048 possible tiles (B <= 512 at S = 50) run a
 *   workgroup per tile -- at most 1024 tiles, at most 3/4 of the CUs' worth of them tiles of sequences longer than 16 rows; larger batches the
 *   looped form (the resident workgroups -- one per CU; two at D = 64, which the caller declares with & 8 -- with the tiles beyond the grid
 *   handed out by a counter: any number of tiles) where it is the faster of the two: at most 1.5 tiles of long sequences and at most 10 tiles
 *   in all per resident workgroup.
 *   & 1: a sequence of 3 - 4 tiles becomes TWO work items (its first two tiles / the rest) that run in two workgroups at
 *   once and hand k, v (forward) and the partial dK, dV (backward) over through the tape -- the launch lasts as long as its largest
 *   item.  Only done when every item of the plan still gets a workgroup of its own (<= ncu items); needs a tape whose flag words
 *   (the tail of re_sasrec_tape_bytes) are zero before the first launch -- every launch leaves them zero -- i.e. training launches. */
size_t re_sasrec_plan_bytes(int64_t B, int64_t S);
int re_sasrec_batch_prep(const int64_t* seq, const int64_t* pos, const int64_t* neg, int64_t B, int64_t S, int32_t ncu,
                         int32_t max_tiles, int32_t split_long, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out, uint8_t* valid, int32_t* count,
                         int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed, int64_t step, double lr,
                         double beta1, double beta2, re_stream_t stream);
/* The same launch with the weight preparation of the D = 64 one-tile-per-workgroup step in extra workgroups: the encoder's matrices
 * (block_params / last_w / last_b as for re_sasrec_encoder_step) as bf16 hi / mid fragment planes into that step's workspace `ws`, the
 * launch epoch in `tape`'s flag area advanced.  The step that follows on the same stream is then called with part + 8
 * (re_sasrec_encoder_step_part) and launches no preparation kernel of its own.  block_params == NULL: no weight preparation.
 * loss_acc != NULL: loss_acc[0] += prev_loss[0] * loss_weight -- the PREVIOUS step's loss folded into an epoch accumulator by a launch
 * that runs anyway (the reference reads loss.item() per step, SASRec/main.py:252-256; an accumulation launch per step costs 4 us). */
int re_sasrec_batch_prep_w(const int64_t* seq, const int64_t* pos, const int64_t* neg, int64_t B, int64_t S, int32_t ncu,
                           int32_t max_tiles, int32_t split_long, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out, uint8_t* valid, int32_t* count,
                           int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed, int64_t step, double lr,
                           double beta1, double beta2, const float* const* block_params, const float* last_w, const float* last_b,
                           int64_t L, int64_t D, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes, const float* prev_loss,
                           float* loss_acc, float loss_weight, re_stream_t stream);

/* SAMPLE + PREPARE in one launch: row b of the batch is user order[b0 + b] of the SASRec training chain -- the rows and draws of
 * re_seq_train_sample(ptr, items, sorted_items, order, n_order, b0, B, S, N, sample_seed, sample_step, ...) (SASRec/main.py:143-157) --
 * written straight into seq_out / pos_out / neg_out (the static buffers a captured step reads; required) together with everything
 * re_sasrec_batch_prep derives from a batch; block_params != NULL: + the tile step's weight fragments as re_sasrec_batch_prep_w.
 * A sampled training step is then ONE preparation launch + the step, not sampler + preparation + step. */
int re_seq_train_sample_prep(const int64_t* ptr, const int64_t* items, const int64_t* sorted_items, const int64_t* order, int64_t n_order,
                             int64_t b0, int64_t N, uint32_t sample_seed, uint32_t sample_step, int64_t* users, int64_t B, int64_t S,
                             int32_t ncu, int32_t max_tiles, int32_t split_long, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out,
                             uint8_t* valid, int32_t* count, int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed,
                             int64_t step, double lr, double beta1, double beta2, const float* const* block_params, const float* last_w,
                             const float* last_b, int64_t L, int64_t D, void* tape, size_t tape_bytes, void* ws, size_t ws_bytes,
                             const float* prev_loss, float* loss_acc, float loss_weight, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K6/K7  fused SASRec encoder (D = 64 or 128, S <= 64, L <= 4, 1 head): one workgroup per work item of the plan, activations in LDS.
 * Replaces the per-block aten chain of SASRec/main.py:163-176 (after_one_block), :31-50 (PointWiseFeedForward),
 * :188-191 (block loop + lastLN): LN -> q/k/v -> causal softmax (pads attended as keys, K/V not layer-normed)
 * -> out_proj + residual -> LN -> Conv1d(k=1) FFN + residual -> pad mask; all five dropout sites in-kernel
 * (counter-based masks, csrc/re_rng.h; drop_p = 0 disables them).
 *   x0 [B,S,D] = re_sasrec_embed output, or NULL: the input rows are built inside the kernel from the item table E [R, D]
 *   (row 0 = padding), the position table Ptab [S, D] and `scale` (= sqrt(D)) -- SASRec/main.py:181-187, same dropout stream as
 *   re_sasrec_embed.  seq [B,S] int64 (0 = pad);  u [B,S,D] = userEmbds (lastLN output).
 *   block_params: HOST array of 12*L DEVICE pointers, per block in this order:
 *     attnLNs.weight, attnLNs.bias, in_proj_weight [3D,D], in_proj_bias [3D], out_proj.weight [D,D], out_proj.bias,
 *     fwdLNs.weight, fwdLNs.bias, conv1.weight [D,D(,1)], conv1.bias, conv2.weight, conv2.bias
 *   plan: from re_sasrec_batch_prep for THIS seq.  ncu: the launch grid (compute units to fill; <= 1024).
 *   tape: NULL for inference; otherwise re_sasrec_tape_bytes() bytes that receive the activations the backward
 *   needs (x, LN_a(x), q, k, v, P, o, x1, LN_f(x1), relu(h), LN statistics), indexed by the plan's compact rows.
 *   fill_pads != 0: the rows of u at the pad positions in front of a sequence are set to lastLN.bias (= what the reference's
 *   encode returns there; inference).  Otherwise they are not written (training: nothing on the path reads them), nor are those of dx0.
 *   D = 128: a work item holds 32 rows in LDS; a longer sequence is taken in chained parts that hand their k, v over through the
 *   tape -- which therefore must be given (also for inference) when S > 32.
 * re_sasrec_encoder_bwd: given dU [B,S,D] (gradient w.r.t. u) and the tape of the SAME (drop_p, seed, plan) forward, ONE launch
 *   for all blocks writes dx0 [B,S,D] and six dY operands per block; the weight gradients dW = dY^T X are then split-K products
 *   over all compact rows (second launch) and a fixed-order reduction (third launch) that OVERWRITES the parameter gradients:
 *   block_grads is a HOST array of 12*L DEVICE pointers in the order above.  Deterministic, no float atomics.
 *   dPtab == NULL: dx0 = gradient w.r.t. x0 (for re_sasrec_embed_bwd).  dPtab != NULL: re_sasrec_embed_bwd is fused in -- dx0
 *   receives the item-gradient contribution rows (pad mask, embedding dropout mask, * scale) and dPtab [S, D] the
 *   position-table gradient.
 *   dU_rows (optional, [re_sasrec_plan_rows, D]): the upstream gradient indexed by the plan's compact rows (re_sasrec_loss_rows);
 *   replaces dU, which may then be NULL.  dx0_rows (optional): receives the rows of dx0 once more, in compact order (region 0 of
 *   re_sasrec_loss_rows' g_rows). */
size_t re_sasrec_tape_bytes(int64_t B, int64_t S, int64_t D, int64_t L);
/* The tape's array offsets in floats (tools / tests that read a tape back; arrays are indexed by the plan's compact rows, those of
 * block l start at l * per_block): out[0..n) = per_block, X, A, Q, K, V, O, X1, Y, HR, P, SA, SF, PP, MK, XL, SL, FLAGS, total. */
int re_sasrec_tape_layout(int64_t B, int64_t S, int64_t D, int64_t L, int64_t* out, int64_t n);
/*
 * K-S  device-side batch assembly: the reference's datapipe chains as one launch per batch, no host work.
 *   The training interactions live in HBM as CSR over users: ptr int64[U + 1], items int64[nnz] in chronological order,
 *   sorted_items int64[nnz] ascending per user (the "seen" probe).  order int64[n_order]: for re_seq_train_sample the epoch's
 *   shuffled list of users with >= 2 training items (`shuffled_seqs_source`), for re_gen_train_sample the users with >= 1 item.
 * re_seq_train_sample: `seq_train_yielding_pos_(1, -1) -> seq_train_sampling_neg_(1) -> add_(1, (ISeq,)) -> lpad_(S, ..., 0)`
 *   (SASRec/main.py:143-157; row contract HSTU/sampler.py:47-125): row b = user order[b0 + b] (rows past n_order are all padding);
 *   seq [B, S] = the last S inputs + 1, pos [B, S] = the items that follow them, neg [B, S] = one uniform item per real position
 *   outside the user's training set, all left-padded with 0; users [B] (optional) = the rows' user ids (-1 on padding rows).
 * re_gen_train_sample: `choiced_user_ids_source -> gen_train_sampling_pos_ -> gen_train_sampling_neg_(1)` (MF-BPR/main.py:60-68):
 *   users [B] uniform (with replacement) from `order`, pos [B] one of the user's training items, neg [B] one unseen item.
 *   Draws are a pure function of (seed, step, row, position): counter-based (csrc/re_rng.h), reproducible, capturable.
 */
int re_seq_train_sample(const int64_t* ptr, const int64_t* items, const int64_t* sorted_items, const int64_t* order, int64_t n_order,
                        int64_t b0, int64_t B, int64_t S, int64_t N, uint32_t seed, uint32_t step, int64_t* users, int64_t* seq,
                        int64_t* pos, int64_t* neg, re_stream_t stream);
int re_gen_train_sample(const int64_t* ptr, const int64_t* items, const int64_t* sorted_items, const int64_t* order, int64_t n_order,
                        int64_t B, int64_t N, uint32_t seed, uint32_t step, int64_t* users, int64_t* pos, int64_t* neg,
                        re_stream_t stream);

/* The pair criteria of re_pair_loss_fwd_bwd on the plan's compact rows only (SASRec/main.py:199-215 without the boolean-mask
 * compaction and without the padding positions): NR = re_sasrec_plan_rows(B, S) bounds the number of rows; per live row r
 * (position gid of the plan's row map):  dU_rows [NR, D] row r = d loss / d u[gid] (zero where seq[gid] == 0),
 * g_rows [3, NR, D]: rows [1][r] = dpl * u, [2][r] = dnl * u (region 0 is re_sasrec_encoder_bwd's dx0_rows),
 * keys int32 [3, NR] = seq[gid] | e_off + pos[gid] | e_off + neg[gid] (0 = no contribution) -- the operands of
 * re_scatter_add_rows_small(g_rows, keys, 3, NR, plan + 4 bytes (the tile count), 16, ...).  loss[0] = mean over the `count[0]`
 * valid positions (count from re_sasrec_batch_prep).  U is userEmbds [B*S, D].  ws: re_sasrec_loss_rows_workspace_bytes() bytes,
 * zero before the first call (every call leaves it zero).  D = 64 or 128; table row 0 is the padding row. */
int64_t re_sasrec_plan_rows(int64_t B, int64_t S);
size_t re_sasrec_loss_rows_workspace_bytes(void);
int re_sasrec_loss_rows(const float* U, const float* E, int64_t R, int64_t D, int64_t e_off, const int64_t* seq, const int64_t* pos,
                        const int64_t* neg, int64_t B, int64_t S, const void* plan, int kind, const int32_t* count, float* loss,
                        float* dU_rows, float* g_rows, int32_t* keys, void* ws, size_t ws_bytes, re_stream_t stream);
/* Training forward + criterion in ONE launch: re_sasrec_encoder_fwd (x0 == NULL: input rows built from the tables; tape required)
 * with re_sasrec_loss_rows folded into every work item's tail -- u is still in LDS, and the gathers of E[pos], E[neg] overlap the
 * last block.  Same outputs as the two calls (u, tape; loss, dU_rows, g_rows[1:3], keys); ws: 256 bytes, zero before the first
 * call (every call leaves it zero). */
int re_sasrec_encoder_fwd_loss(const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, const int64_t* pos,
                               const int64_t* neg, int64_t B, int64_t S, int64_t D, int64_t L, const float* const* block_params,
                               const float* last_w, const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                               const void* plan, int32_t ncu, float* u, void* tape, size_t tape_bytes, int64_t e_off, int kind,
                               const int32_t* count, float* loss, float* dU_rows, float* g_rows, int32_t* keys, void* ws,
                               size_t ws_bytes, re_stream_t stream);
int re_sasrec_encoder_fwd(const float* x0, const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, int64_t B,
                          int64_t S, int64_t D, int64_t L, const float* const* block_params, const float* last_w,
                          const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev, const void* plan, int32_t ncu,
                          float* u, void* tape, size_t tape_bytes, int32_t fill_pads, re_stream_t stream);
/* The encoder's whole training step: re_sasrec_encoder_fwd_loss followed by re_sasrec_encoder_bwd(dU_rows = the head's rows,
 * dx0_rows = g_rows region 0, dPtab given), with the two item kernels as ONE launch (per work item: forward, criterion, backward)
 * + the weight-gradient and reduction launches.  Same results, bit for bit.  ws as re_sasrec_encoder_bwd; loss_ws as
 * re_sasrec_encoder_fwd_loss.
 *   D = 64 runs ONE TILE PER WORKGROUP, four waves per tile (csrc/enc_tile.hip: a tile's activations in registers, one 16-feature
 *   strip per wave, bf16 hi / mid split products on the XDL pipe -- results within the 1e-4 bound of the fp32 kernels', not their
 *   bits) whenever the plan says so (re_sasrec_batch_prep decides per batch where it is the faster of the two -- see split_long there;
 *   one workgroup per CU, each taking its block index's tile and then tiles from a counter), and the workgroup-per-item kernel
 *   otherwise; both are enqueued, one of them returns at once.  The tape's flag words must be zero before the first launch. */
int re_sasrec_encoder_step(const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, const int64_t* pos,
                           const int64_t* neg, int64_t B, int64_t S, int64_t D, int64_t L, const float* const* block_params,
                           const float* last_w, const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                           const void* plan, int32_t ncu, float* u, void* tape, size_t tape_bytes, int64_t e_off, int kind,
                           const int32_t* count, float* loss, float* dU_rows, float* g_rows, int32_t* keys, void* loss_ws,
                           size_t loss_ws_bytes, float* dx0, float* dPtab, float* const* block_grads, float* g_last_w,
                           float* g_last_b, void* ws, size_t ws_bytes, re_stream_t stream);
/* The same step in parts.  `part` is a mask of what to launch: 1 = the one-tile-per-workgroup kernels (D = 64), 2 = the
 * workgroup-per-item kernel (exactly one of the two does the work: the plan decides on the device, the other returns at once),
 * 4 = the weight gradients (from the tape the item kernels left); 0 = 7 = everything.  + 8: the weight fragments were prepared by
 * re_sasrec_batch_prep_w for this step.  The two item kernels are independent of each other and the item table's scatter-add
 * depends on them alone: a caller runs 1 and 2, then 4 and the scatter-add, as parallel branches on two streams
 * (recboard_amd/sasrec.py does, inside the captured step) and joins them before the optimizer.
 * adam != NULL (with part & 4): the reduction that finishes the encoder's gradients (position table, every block's matrices and vectors,
 * lastLN) also applies their dense Adam update -- all of them must then be views of one arena (adam->grad_base). */
int re_sasrec_encoder_step_part(const float* E, int64_t R, const float* Ptab, float scale, const int64_t* seq, const int64_t* pos,
                                const int64_t* neg, int64_t B, int64_t S, int64_t D, int64_t L, const float* const* block_params,
                                const float* last_w, const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                                const void* plan, int32_t ncu, float* u, void* tape, size_t tape_bytes, int64_t e_off, int kind,
                                const int32_t* count, float* loss, float* dU_rows, float* g_rows, int32_t* keys, void* loss_ws,
                                size_t loss_ws_bytes, float* dx0, float* dPtab, float* const* block_grads, float* g_last_w,
                                float* g_last_b, void* ws, size_t ws_bytes, int32_t part, const re_adam_fuse* adam, re_stream_t stream);
size_t re_sasrec_encoder_bwd_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t L);
/* Where the library places the regions of that workspace in a buffer at address `base` (the address is not dereferenced; the two aligned
 * regions depend on it): out[0..6] = byte offsets of the workgroup slabs, the matrices' split-K partials, the position table's group partials,
 * the gradient tape, the tile kernels' weight fragments, the tiles' dK / dV inboxes, and the first byte behind the last region
 * (<= re_sasrec_encoder_bwd_workspace_bytes for every base).  Host only; for layout checks (tests/test_workspace_layout.py). */
int re_sasrec_encoder_bwd_workspace_layout(int64_t B, int64_t S, int64_t D, int64_t L, uint64_t base, uint64_t* out);
int re_sasrec_encoder_bwd(const float* dU, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                          const float* const* block_params, const float* last_w, const float* last_b, float drop_p, uint32_t seed,
                          const uint32_t* seed_dev, const void* tape, const void* plan, int32_t ncu, float scale, float* dx0,
                          float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b, const float* dU_rows,
                          float* dx0_rows, void* ws, size_t ws_bytes, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Ranking metrics from the sorted top-K list of re_score_topk (freerec.metrics via Coach.evaluate, contract mirrored at
 * UniSRec/main.py:428-447).  topk_idx [B, Kmax] int64; targets as CSR (tgt_ptr[B+1], tgt_idx) of item ids;
 * h_ks = HOST array of nk <= 8 cut-offs (each <= Kmax <= 64).
 * per_user [B, nk, 5] = (HITRATE, PRECISION, RECALL, NDCG, MRR) @ k;  sums [nk*5] (optional) = totals over the B users
 * in a fixed order (the caller divides by the number of users, as Coach.monitor does with n = batch size). */
int re_rank_metrics(const int64_t* topk_idx, int64_t B, int64_t Kmax, const int64_t* tgt_ptr, const int64_t* tgt_idx,
                    const int32_t* h_ks, int32_t nk, float* per_user, float* sums, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K8  CSR SpMM for LightGCN propagation:  Y = A X (+ beta Z);  if ACC: ACC += acc_scale * Y.
 * Replaces `self.Adj @ allEmbds` + `avgEmbds += allEmbds / (L+1)` (LightGCN/main.py:81-84) and, Adj being symmetric,
 * the transposed product of the backward pass.  crow int64[nrows+1], col int64[nnz], val f32[nnz]; X [ncols, D],
 * Y/Z/ACC [nrows, D]; Y must not alias X.  Plan (computed once per adjacency, all device arrays): row_order[nrows] =
 * row ids in descending-degree order (NULL = natural order, then nlong must be 0); its first nlong rows (more than 512
 * non-zeros) are cut into chunks of 2048 non-zeros: chunk_ptr[nlong+1] = first chunk of each long row,
 * chunk_row[nchunks] = index (into row_order) of the row a chunk belongs to; ws >= nchunks*D*4 bytes holds the chunk
 * partials.  Fixed summation order => bitwise reproducible.
 * re_rows_sqnorm: out[0] (+)= scale * sum_i ||W[idx[i],:]||^2  -- `criterion.regularize(rows, "l2")`
 * (LightGCN/main.py:99-106) with scale = 1/2 / B. */
int re_spmm_csr(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                const int64_t* row_order, int64_t nlong, const int32_t* chunk_row, const int64_t* chunk_ptr,
                int64_t nchunks, const float* X, int64_t D, float* Y, const float* Z, float beta, float* ACC,
                float acc_scale, void* ws, size_t ws_bytes, re_stream_t stream);
/* re_spmm_csr with the plan's TWO ROW CLASSES kept apart by XCD and / or the once-read streams moved non-temporally.  A bipartite
 * adjacency's user rows gather item rows and its item rows gather user rows: row_order = [nlong long rows | class 0 | class 1], `split` =
 * position of class 1's first row in row_order (<= nlong or == nrows: one class, as re_spmm_csr), `xcd_share` (1 .. 7) = how many of the 8
 * XCD labels (blockIdx % 8) walk class 0 -- each XCD's L2 then holds the hot rows of ONE part of X.  flags & 1: (col, val), Z, Y and ACC
 * go through the caches with the non-temporal hint.  flags & 2: the long rows' chunk partials are added INSIDE the launch (by the workgroup
 * that brings a row's last chunk, in chunk order) instead of by a second launch: `ws` then holds, behind the partials rounded up to 16 bytes,
 * nlong int32 arrival counters -- ws_bytes >= align16(nchunks * D * 4) + nlong * 4, the counters ZERO before the first call (every call
 * leaves them zero).  flags & 4 (nrows == ncols): ACC = acc_scale * (X[row] + Y[row]) instead of ACC += acc_scale * Y[row] -- the running
 * mean of LightGCN's layers (LightGCN/main.py:77-86) started by the first propagation itself.  Results are bit-identical to re_spmm_csr's (a
 * row's sum does not depend on who computes it). */
int re_spmm_csr_split(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                      const int64_t* row_order, int64_t nlong, int64_t split, int32_t xcd_share, int32_t flags,
                      const int32_t* chunk_row, const int64_t* chunk_ptr, int64_t nchunks, const float* X, int64_t D, float* Y,
                      const float* Z, float beta, float* ACC, float acc_scale, void* ws, size_t ws_bytes, re_stream_t stream);
/* re_spmm_csr_split with a bit per row of X (src_mask[i >> 5] bit i & 31; NULL: none): 0 = that row of X is all zeros and is not fetched --
 * the product of a matrix with mostly-zero rows (the first propagation of LightGCN's backward pass: the scatter of 3 B gradient rows into
 * 122 915) gathers only what can contribute.  Adding exact zeros changes no sum: results are the unmasked ones bit for bit, and the masked-out
 * rows are never read: the caller may leave them unwritten (re_scatter_apply accumulate = 2).  re_row_mask builds the mask from the scatter's
 * row list.
 * flags & 8: the mask also names Z's non-zero rows (a row of Z with a zero bit is not read); flags & 16: the mask is for Z only.
 * row_ptrs (optional, [nrows][2]): crow[row_order[i]], crow[row_order[i] + 1] -- the row pointers in WALKING order, fetched beside the row id
 * instead of behind it (one dependent memory round trip less per row). */
int re_spmm_csr_masked(const int64_t* crow, const int64_t* col, const float* val, int64_t nrows, int64_t ncols,
                       const int64_t* row_order, int64_t nlong, int64_t split, int32_t xcd_share, int32_t flags,
                       const int32_t* chunk_row, const int64_t* chunk_ptr, int64_t nchunks, const float* X, int64_t D, float* Y,
                       const float* Z, float beta, float* ACC, float acc_scale, const uint32_t* src_mask, const int64_t* row_ptrs, void* ws,
                       size_t ws_bytes, re_stream_t stream);
int re_row_mask(const int64_t* rows, int64_t n, int64_t nbits, uint32_t* mask, re_stream_t stream);
size_t re_rows_sqnorm_workspace_bytes(void);
int re_rows_sqnorm(const float* W, int64_t R, int64_t D, const int64_t* idx, int64_t n, float scale, float* out,
                   int accumulate, void* ws, size_t ws_bytes, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K9  DeepFM multi-field embedding bag (DeepFM/main.py:58-62, :80-85, :204-209).  The F per-field tables are ONE
 * concatenated table T [rows_total, D] (+ LR vector TL [rows_total]); offsets[F] (device) = first row of each field.
 *   fwd: E[b,f,:] = T[offsets[f] + x[b,f], :]  (written as [B, F*D], the MLP input)
 *        fm_lr[b] = 0.5 * sum_d((sum_f E)^2 - sum_f E^2) + sum_f TL[...] + lr_bias[0]
 *   bwd: gE[b,f,:] = dE_mlp[b,f,:] + dlogit[b] * (sum_f' E[b,f',:] - E[b,f,:]);  gL[b,f] = dlogit[b]
 *        (contribution rows for re_scatter_add_rows with indices offsets[f] + x[b,f]; dE_mlp may be NULL)
 * F <= 64, D <= 16 (the reference uses D = 10).
 * re_bce_logits: loss[0] = mean BCE-with-logits (DeepFM/main.py:214), dlogit[i] = (sigmoid(x_i) - y_i) / n,
 * dsum[0] (optional) = sum_i dlogit[i] (gradient of the LR bias).  labels are fp32 0/1. */
int re_fm_bag_fwd(const float* T, const float* TL, const float* lr_bias, const int64_t* offsets, int64_t rows_total,
                  const int64_t* x, int64_t B, int64_t F, int64_t D, float* E, float* fm_lr, int64_t* rows_out, int32_t* keys_t,
                  re_stream_t stream);
/* (rows_out [B * F], optional: offsets[f] + x[b, f] -- the destination rows of the backward's scatter-add, re_scatter_plan's `idx`;
 *  keys_t [F * B] int32, optional: x[b, f] field-major -- re_fm_table_grad's keys) */
/* re_fm_table_grad: the two tables' gradients from re_fm_bag_bwd's contribution rows in ONE launch.  keys_t[f * B + b] = x[b, f] (field-major
 * int32: re_fm_bag_fwd's keys_t); contribution (b, f) goes to row offsets[f] + x[b, f].  The work is cut by destination rows: `slices` (device, int32 [n_slices][4]: field, first row, end row -- both
 * relative to offsets[field] -- and a spare word) must cover every row that can occur, each exactly once; a workgroup per slice collects the
 * field's keys that fall into it and sums them per row:
 *   gT[r, :] = sum over (b, f) with offsets[f] + x[b, f] = r of gE[b, f, :],   gTL[r] = the same of gL[b, f]      (a fixed order: reproducible)
 * Any slicing gives a correct result; ~40 expected keys a slice (64 or fewer are one wave's work in registers) is the fast one.  Rows nobody
 * refers to are NOT written: zero-fill gT [rows_total, D] and gTL [rows_total] first.  Keys outside every slice are dropped.
 * B <= 8192, F <= 64, D <= 15, rows of a field < 2^19 - 1, else RE_EUNSUPPORTED (use re_scatter_add_rows). */
int re_fm_table_grad(const int32_t* keys_t, int64_t B, int64_t F, const int64_t* offsets, int64_t rows_total, const int32_t* slices,
                     int64_t n_slices, const float* gE, const float* gL, int64_t D, float* gT, float* gTL, re_stream_t stream);
int re_fm_bag_bwd(const float* E, const float* dE_mlp, const float* dlogit, int64_t B, int64_t F, int64_t D, float* gE,
                  float* gL, re_stream_t stream);
int re_bce_logits(const float* logits, const float* labels, int64_t n, float* loss, float* dlogit, float* dsum,
                  re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * K10  dense Adam with coupled L2 (torch.optim.Adam semantics, eps 1e-8, no amsgrad), one launch over a flat
 * parameter arena.  Replaces `self.optimizer.step()` (SASRec/main.py:250; cfg dump
 * benchmark/Amazon2014Beauty_550_LOU/SASRec.json:254-300).  step is 1-based.  Hyper-parameters are doubles because
 * torch derives 1-beta and the bias corrections in double precision before rounding to fp32. */
/* AUC of a prediction model's scores (DeepFM/configs/Frappe_x1_BARS.yaml:101-102 `monitors: [LOGLOSS, AUC]`; scores =
 * recommend_from_pool outputs, DeepFM/main.py:217-219; labels > 0.5 = positive): the Mann-Whitney statistic counted pairwise --
 * no sort, integer counts (exact, deterministic), ties count one half.  auc[0] = 0.5 if a class is empty.  ws: 256 bytes. */
/* Pool ranking (`--ranking=pool`; recommend_from_pool: SASRec/main.py:230-236, MF-BPR/main.py:106-109, LightGCN/main.py:122-125).
 * re_score_pool: out[b][p] = <Q[b,:], E[pool[b][p],:]> (natural-k fmaf chain: the value re_score_dense gives the pair, bit for bit);
 *   pool int64 [B, P] item ids; an id outside [0, N) scores -inf.  D a multiple of 4, rows 16-byte aligned.
 * re_pool_topk: per row of scores [B, P] the exact top-K (vals descending, idx = position in the row, ties -> lowest position; slots
 *   beyond P get (-inf, -1)); P <= 1024.  With the evaluation pipes' layout (target at position 0) re_rank_metrics on idx against
 *   the target list {0} gives every NAME@k. */
int re_score_pool(const float* Q, const float* E, const int64_t* pool, int64_t B, int64_t P, int64_t N, int64_t D, float* out,
                  re_stream_t stream);
int re_pool_topk(const float* scores, int64_t B, int64_t P, int64_t K, float* vals, int64_t* idx, re_stream_t stream);
size_t re_auc_workspace_bytes(void);
int re_auc(const float* scores, const float* labels, int64_t n, float* auc, void* ws, size_t ws_bytes, re_stream_t stream);
/* hipGraph-friendly variant of re_adam_step: hyper (DEVICE float[2]) = { lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t) };
 * n must be a multiple of 4. */
/* The per-step device words of a captured step, written by one tiny launch (kernel arguments: no host buffer to keep alive):
 * state[0] = seed, state[1] = 0, state[2..3] = bits of { lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step) } -- the layout
 * re_sasrec_batch_prep writes; state doubles as `seed_dev` of the dropout entry points and, from word 2, as `hyper` of re_adam_step_dev. */
int re_step_state(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, re_stream_t stream);
/* re_step_stage_inputs: re_step_state (state may be NULL: no scalars) and up to 8 transfers into a captured step's static buffers in ONE launch:
 * segment i writes bytes[i] bytes at dst[i] -- kind 0: copied from src[i]; kind 1: four-byte floats converted from the int64 words at src[i]
 * (labels); kind 2: zeros (src[i] ignored).  Segments must not overlap. */
int re_step_stage_inputs(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, int32_t n, void* const* dst,
                         const void* const* src, const int64_t* bytes, const int32_t* kind, re_stream_t stream);
int re_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, double beta1,
                     double beta2, double eps, double weight_decay, re_stream_t stream);
/* clip_grad_norm_ + Adam (DeepFM/main.py:264-268).  re_grad_clip_coef: coef_norm[0] = min(1, max_norm / (||g||_2 + 1e-6)) (torch's
 * formula), coef_norm[1] = ||g||_2 over the flat gradient, as device words (deterministic two-stage sum).  re_adam_step_scaled: Adam on
 * gscale[0] * g with the scaled gradient written back to g; step >= 1: the host's bias corrections, step == 0: the device words `hyper`
 * (re_adam_step_dev's).  n a multiple of 4, 16-byte aligned buffers. */
size_t re_grad_clip_workspace_bytes(void);
int re_grad_clip_coef(const float* g, int64_t n, float max_norm, float* coef_norm, void* ws, size_t ws_bytes, re_stream_t stream);
int re_adam_step_scaled(float* p, float* g, float* m, float* v, int64_t n, int64_t step, double lr, const float* hyper, double beta1,
                        double beta2, double eps, double weight_decay, const float* gscale, re_stream_t stream);
/* re_adam_step_clip2: clip_grad_norm_(.., max_norm) and Adam over an arena of two weight-decay groups -- [0, n_first) with wd_first, the rest with
 * wd_rest (DeepFM/main.py:187-199, 267-268) -- in two launches (square-norm partials; then every workgroup forms the coefficient
 * min(1, max_norm / (||g|| + 1e-6)) from them, scales g on the way through -- written back -- and updates).  coef_norm (optional) receives
 * [coefficient, ||g||].  n, n_first multiples of 4, 16-byte aligned pointers; step >= 1: host-side bias corrections, step == 0: hyper (device
 * float[2]).  ws: re_grad_clip_workspace_bytes(). */
int re_adam_step_clip2(float* p, float* g, float* m, float* v, int64_t n, int64_t n_first, int64_t step, double lr, const float* hyper,
                       double beta1, double beta2, double eps, double wd_first, double wd_rest, float max_norm, float* coef_norm, void* ws,
                       size_t ws_bytes, re_stream_t stream);
/* The owner's half of a data-parallel step (N > 1 replicas of one flat parameter arena; the reference trains one replica:
 * freerec/launcher.py's Coach, SASRec/main.py:264-275): this rank owns p[0 .. n) (a slice of the arena); parts + r * part_stride (r < nparts)
 * is rank r's gradient for the slice (what an all-to-all of the ranks' gradient arenas delivers).  g = gscale (((part 0 + part 1) + ...)
 * in rank order); g_out (optional) receives it; then re_adam_step's update on (p, m, v).  n, part_stride multiples of 4, 16-byte aligned. */
int re_adam_step_reduce(float* p, const float* parts, int nparts, int64_t part_stride, float* g_out, float* m, float* v, int64_t n,
                        int64_t step, double lr, const float* hyper, double beta1, double beta2, double eps, double weight_decay,
                        double gscale, re_stream_t stream);
int re_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, double lr,
                 double beta1, double beta2, double eps, double weight_decay, re_stream_t stream);

/* Cross entropy over materialised logits [M, N] (row stride ld), forward and backward IN PLACE (SASRec --loss CE,
 * SASRec/main.py:217-219; CrossEntropy4Logits(mean) = F.cross_entropy): row_loss[m] = logsumexp(x_m) - x_m[labels[m]],
 * loss[0] = mean(row_loss), and logits <- (softmax(x_m) - onehot(labels[m])) / M, the gradient w.r.t. the logits. */
int re_ce_rows(float* logits, int64_t M, int64_t N, int64_t ld, const int64_t* labels, float* row_loss, float* loss,
               re_stream_t stream);
/* The same cross entropy without the [M, N] matrix (K5, SASRec/main.py:217-219 at catalog sizes where M x N does not fit or is not
 * worth its bytes): the caller walks the catalog in column chunks [col0, col0 + Nc) and materialises ONE chunk of logits at a time.
 *   re_ce_chunk_stats  folds a chunk into the running row statistics (online log-sum-exp: rowmax, rowsum = sum exp(x - rowmax)) and
 *                      records tgt[m] = x[m, labels[m] - col0] when the label lies in the chunk; first != 0 starts the statistics.
 *   re_ce_chunk_loss   after the last chunk: row_loss[m] = log(rowsum) + rowmax - tgt, loss[0] = mean.
 *   re_ce_chunk_grad   rewrites a RECOMPUTED chunk in place to (softmax - onehot) / M, the gradient w.r.t. that chunk's logits. */
int re_ce_chunk_stats(const float* logits, int64_t M, int64_t Nc, int64_t ld, int64_t col0, const int64_t* labels, int first,
                      float* rowmax, float* rowsum, float* tgt, re_stream_t stream);
int re_ce_chunk_loss(const float* rowmax, const float* rowsum, const float* tgt, const int64_t* labels, int64_t M, int64_t N,
                     float* row_loss, float* loss, re_stream_t stream);
int re_ce_chunk_grad(float* logits, int64_t M, int64_t Nc, int64_t ld, int64_t col0, const int64_t* labels, const float* rowmax,
                     const float* rowsum, re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * General fp32 GEMM on the matrix cores: C[M,N] = alpha * op(A)[M,K] op(B)[K,N] + beta * C (+ bias[n]) (+ ReLU).
 * transX = 0: the operand is stored as op(X) row-major (A: [M,K] lda, B: [K,N] ldb); transX = 1: stored transposed
 * (A: [K,M] lda, B: [N,K] ldb).  `y = x W^T + b` (nn.Linear, DeepFM/main.py:112-121; SASRec CE logits
 * SASRec/main.py:217) is transA = 0, transB = 1.  Exact fp32 (k-ordered fmaf chains); long-K skinny products are split
 * along K into slabs reduced in a fixed order (workspace from re_gemm_f32_workspace_bytes). */
size_t re_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K);
int re_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* bias, int relu, void* ws,
                size_t ws_bytes, re_stream_t stream);
/* re_gemm_f32_slabs: re_gemm_f32 (beta = 0, no bias / relu) whose split-K reduction is left to the caller -- *nsplit_out > 1: the partial
 * products are in ws ([nsplit][M][N], alpha not applied) and C is not written yet; == 1: C holds the product.  re_gemm_splitk_reduce_many
 * finishes up to 8 such products in one launch (the slab-order sums of the one-product path: the same bits): DeepFM's three weight-gradient
 * products of a step (DeepFM/main.py:103-124 backward). */
int re_gemm_f32_slabs(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, void* ws, size_t ws_bytes, int32_t* nsplit_out, re_stream_t stream);
int re_gemm_splitk_reduce_many(int32_t n, const float* const* slabs, const int32_t* nsplit, const int64_t* M, const int64_t* N,
                               const float* alpha, float* const* C, const int64_t* ldc, re_stream_t stream);
/* C = alpha op(A) op(B) + bias, and in the same launch the BatchNorm batch statistics of C's columns as per-64-row partials:
 * colstats [M / 64][2][N] = (mean, M2 = sum of squared deviations from it) of rows [64 b, 64 b + 64) -- `bn(linear(x))`
 * (DeepFM/main.py:119-124) without a second pass over the linear map's output; re_bn_relu_drop_fwd_pre takes them.
 * M a multiple of 64, operands 16-byte aligned with leading / contiguous dimensions in multiples of 4; otherwise
 * RE_EUNSUPPORTED (run re_gemm_f32 + re_bn_relu_drop_fwd instead). */
int re_gemm_f32_colstats(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                         const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float* colstats,
                         re_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * DeepFM MLPBlock pieces around re_gemm_f32 (DeepFM/main.py:103-124: Linear -> BatchNorm1d -> ReLU -> Dropout):
 * re_bn_relu_drop_fwd: a = dropout(relu(bn(z))) over z [M, N]; gamma == NULL means "no BatchNorm" (a = dropout(relu(z))).
 *   training != 0: batch statistics (biased variance), running_mean/var updated with `momentum` (unbiased variance), the
 *   engine's counter-based dropout mask (element id m*N+n, stream_id); otherwise running statistics, no dropout.
 *   stats [2N] receives (mean, rstd) for the backward.
 * re_bn_relu_drop_bwd: given da, writes dz [M, N] (gradient w.r.t. z), dgamma [N], dbeta [N] (for gamma == NULL dbeta is
 *   the column sum of dz, i.e. the Linear bias gradient).
 * re_colsum: out[n] = sum_m x[m, n] (bias gradients).  All column reductions are fixed-order (deterministic). */
int re_bn_relu_drop_fwd(const float* z, int64_t M, int64_t N, const float* gamma, const float* beta, float* run_mean,
                        float* run_var, int training, float eps, float momentum, float drop_p, uint32_t seed,
                        const uint32_t* seed_dev /* non-NULL: the seed is read from this device word (captured steps) */,
                        uint32_t stream_id, float* stats, float* a, void* ws, size_t ws_bytes, re_stream_t stream);
int re_bn_relu_drop_bwd(const float* da, const float* a, const float* z, int64_t M, int64_t N, const float* gamma,
                        const float* stats, float drop_p, float* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                        re_stream_t stream);
/* re_bn_relu_drop_fwd in training mode with BatchNorm, the statistics' partials given: colstats [chunks][2][N] (chunk b = rows
 * [b ceil(M / chunks), ...): (mean, M2)) -- what re_gemm_f32_colstats writes with chunks = M / 64.  One launch: every workgroup merges the partials of its 64 columns. */
int re_bn_relu_drop_fwd_pre(const float* z, int64_t M, int64_t N, const float* gamma, const float* beta, float* run_mean,
                            float* run_var, float eps, float momentum, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                            uint32_t stream_id, float* stats, float* a, const float* colstats, int chunks, re_stream_t stream);
int re_colsum(const float* x, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes, re_stream_t stream);
size_t re_mlp_workspace_bytes(int64_t N);
/* DeepFM's last layer + criterion (DeepFM/main.py:151-164 `dnn` ends in Linear(., 1); :201-215 logits = lr + fm + dnn, BCELoss4Logits):
 * re_mlp_head_fwd: logits[m] = fm_lr[m] + <h[m, :], w> + b[0] (fm_lr may be NULL); with labels also loss[0] = mean BCE-with-logits,
 *   dlogit[m] = (sigmoid(logit) - y) / M, dsum[0] = dsum2[0] = sum dlogit (either may be NULL).  re_mlp_head_bwd: da[m, k] = dlogit[m] w[k], dW[k] = sum_m dlogit[m] h[m, k].
 * h [M, K] contiguous, K a multiple of 4, 16-byte aligned (else RE_EUNSUPPORTED: use re_gemm_f32 + re_bce_logits); fixed-order sums. */
size_t re_mlp_head_workspace_bytes(int64_t M, int64_t K);
int re_mlp_head_fwd(const float* h, int64_t M, int64_t K, const float* w, const float* b, const float* fm_lr, const float* labels,
                    float* logits, float* loss, float* dlogit, float* dsum, float* dsum2, void* ws, size_t ws_bytes, re_stream_t stream);
int re_mlp_head_bwd(const float* dlogit, const float* h, const float* w, int64_t M, int64_t K, float* da, float* dW, void* ws,
                    size_t ws_bytes, re_stream_t stream);   /* scratch of the three entry points above (per-chunk column partials) */
/* The backward of dropout(relu(bn(z))) (DeepFM/main.py:119-124) split between the launch that PRODUCES the incoming gradient and one pass:
 *   re_gemm_f32_gated: C = g = act > 0 ? drop_scale alpha op(A) op(B) : 0 (act = the block's output, same leading dimension as C and z) and
 *     part [M / 64][2][N] = per-64-row (sum g, sum g xhat), xhat = (z - stats[n]) stats[N + n].  M a multiple of 64, operands 16-byte aligned
 *     with leading dimensions in multiples of 4, else RE_EUNSUPPORTED (then: re_gemm_f32 + re_bn_relu_drop_bwd).
 *   re_mlp_head_bwd_gated: the same for the gradient that comes from the last Linear(., 1): g [M, K] = h > 0 ? dlogit[m] w[k] / (1 - drop_p) : 0,
 *     part [*chunks_out][3][K] = (sum g, sum g xhat, sum dlogit h); part >= re_mlp_head_workspace_bytes(M, K).
 *   re_bn_bwd_apply: dbeta = sum g, dgamma = sum g xhat (the chunks of `part` [chunks][pstride][N] added in a fixed order), g -> dz in place
 *     = rstd gamma (g - dbeta / M - xhat dgamma / M); extra_out [N] (pstride 3, optional) = the third sums (the last layer's weight gradient). */
int re_gemm_f32_gated(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                      int64_t ldb, float* C, int64_t ldc, const float* act, const float* z, const float* stats, float drop_scale, float* part,
                      re_stream_t stream);
/* (head_ws, loss, dsum, dsum2: optional -- the workspace of a re_mlp_head_fwd call made with labels and loss == NULL, whose per-workgroup
 *  (loss, sum dlogit) partials this launch then adds in order: the criterion's second launch is saved.  NULL: nothing of the kind.) */
int re_mlp_head_bwd_gated(const float* dlogit, const float* h, const float* w, int64_t M, int64_t K, const float* z, const float* stats,
                          float drop_p, float* g, float* part, size_t part_bytes, int* chunks_out, const void* head_ws, float* loss,
                          float* dsum, float* dsum2, re_stream_t stream);
int re_bn_bwd_apply(float* g, const float* z, int64_t M, int64_t N, const float* gamma, const float* stats, const float* part, int chunks,
                    int pstride, float* dgamma, float* dbeta, float* extra_out, re_stream_t stream);

/* dst[i] = alpha * src[i]  (LightGCN/main.py:80 `avgEmbds = allEmbds / (L+1)`) */
int re_scale_copy(float* dst, const float* src, float alpha, int64_t n, re_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* RECENGINE_H */
