// K6/K7 forward, the work of ONE item (device function; kernels: enc_fwd.hip, enc_step.hip): the whole SASRec encoder (embedding front end, all blocks, lastLN) for one work item per workgroup,
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
#pragma once
#include <math.h>

#include "enc_common.h"

// Weight fragments.  D = 64: three register sets, each re-requested as soon as its product is done -- two or more phases before
// its next use (FWREQ; FWUSE is empty).  D = 128: a fragment is 32 registers and three sets in flight spill; it is loaded where it
// is used instead (FWUSE; FWREQ is empty) -- the product behind it is four times longer, the exposed round trip matters less.
#define FWREQ(reg, ptr) do { if (D == 64) wfrag_t<D>(reg, ptr, strip, lane); } while (0)
#define FWUSE(reg, ptr) do { if (D != 64) wfrag_t<D>(reg, ptr, strip, lane); } while (0)

struct SeEmbed {
    const float *E, *P;   // item table [R, D] (row 0 = padding), position table [S, D]; E == nullptr: x0 is given
    int64_t R;
    float scale;
};

// The loss head (optional, training): the pair criteria of SASRec/main.py:199-215 on the item's rows while u is still in LDS --
// what re_sasrec_loss_rows (enc_head.hip) does in a launch of its own.  E == nullptr: no head.
struct EncHead {
    const float* E;               // item table [R, D]
    int64_t R, e_off;
    const int64_t *pos, *neg;     // [B, S]
    int kind;                     // RE_LOSS_BCE / RE_LOSS_BPR
    const int32_t* count;         // number of valid positions (re_sasrec_batch_prep)
    float* loss;                  // [1]
    float* dU_rows;               // [NR, D]
    float* g_rows;                // [3, NR, D]: regions 1, 2 written here
    int32_t* keys;                // [3, NR]
    unsigned long long* acc;      // two zeroed 64-bit words: loss sum in 2^-30 units; arrivals | non-finite partials << 32
};

template <int D, bool TRAIN, bool HEAD>
__device__ __forceinline__ void enc_fwd_item(const float* __restrict__ x0, const SeEmbed& em, const int64_t* __restrict__ seq, int B, int S, int L,
                                             const SasrecParams& P, float drop_scale, uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                             float* __restrict__ tape, const EncTape& T, const EncPlan& PL, int fill_pads, const EncHead& H,
                                             float* lds, int wi, int k, const EncHandoff* HO = nullptr) {
    using C = EC<D>;
    constexpr int KPT = C::KPT;             // keys per thread in the softmax phase
    (void)k;
    float* bX = lds;
    float* bA = bX + C::BUF;
    float* bQ = bA + C::BUF;
    float* bK = bQ + C::BUF;
    float* bV = bK + C::BUF;
    float* sP = bV + C::BUF;
    float* bK0 = sP + C::PBUF;                 // prefix k / v tiles (32 rows) of a chained part / the second half of a split sequence
    float* bV0 = bK0 + C::PRE;
    __shared__ int o_gid[C::ROWS], o_first[C::ROWS], o_pad[C::ROWS], o_sid[C::ROWS], o_start[C::ROWS];
    int *s_gid = HO ? HO->gid : o_gid, *s_first = HO ? HO->first : o_first, *s_pad = HO ? HO->pad : o_pad, *s_sid = HO ? HO->sid : o_sid,
        *s_start = HO ? HO->start : o_start;   // (handed to the backward of the same launch: enc_step_k)
    __shared__ float s_w[C::ROWS];
    __shared__ int s_item[HEAD ? C::ROWS : 1], s_pos[HEAD ? C::ROWS : 1], s_neg[HEAD ? C::ROWS : 1];   // loss head: table rows of the item's rows (0 = none)
    __shared__ float s_red[HEAD ? C::NW : 1];
    float head_sum = 0.f;   // this thread's share of the item's loss (row leaders)
    (void)head_sum;
    __shared__ float s_par[2 * EP_NPAR * D], s_last[2 * D];

    const int tid0 = threadIdx.x;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    {
        int tid = tid0;
        asm volatile("" : "+v"(tid));   // keep per-thread addresses loop-variant (hoisting them out costs registers, then spills)
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: the wave's tile choices are scalar branches)
        const int strip = wave % C::NS, wr = wave / C::NS;
        const int g = lane >> 4, c = lane & 15, col = 16 * strip + c;
        const int r_e = tid / C::TPR, c0_e = (tid % C::TPR) * C::CPT, j0_e = (tid % C::TPR) * KPT;
        const bool row_lead = (tid % C::TPR) == 0;
        (void)g;
        const EncItem whole = enc_item(PL, wi);
        // A sequence with more rows than the LDS holds (MAXT tiles; only at D = 128) is taken in CHAINED parts: the first MAXT tiles
        // as an item of their own, then the later rows with the earlier ones as PREFIX key tiles -- their k, v of every block are on
        // the tape, written by this same workgroup a moment ago (causality: the earlier rows never depend on the later ones).
        // (whole long items of more than MAXT tiles exist only at D = 128; the halves of a SPLIT sequence -- kinds 2, 3 -- are items of
        //  their own in two workgroups and hand k, v over through the tape under flags)
        const int nsub = C::MAXT < 4 ? (whole.nt + C::MAXT - 1) / C::MAXT : 1;
        const bool split_lo = whole.kind == 2, split_hi = whole.kind == 3;
        float* tflags = TRAIN ? tape + T.off_FLAGS : nullptr;
        const int64_t ferr = enc_plan_max_tiles(B, S) * EP_FLAG_WORDS;
        for (int hs = 0; hs < nsub; ++hs) {
        const EncItem it = EncItem{whole.tile0 + hs * C::MAXT, whole.nt - hs * C::MAXT < C::MAXT ? whole.nt - hs * C::MAXT : C::MAXT, whole.kind};
        const int npre = split_hi ? 2 : (C::MAXT < 4 ? hs * C::MAXT : 0);     // prefix key tiles
        const int64_t prow0 = (int64_t)(whole.tile0 - (split_hi ? 2 : 0)) * 16;   // compact row of the sequence's first row
        if (hs > 0) re_sync_full();                          // (a full barrier: the previous part's tape stores have completed)
        const int nrows = 16 * it.nt;
        const int64_t row0 = (int64_t)it.tile0 * 16;
        int mk = 0; (void)mk;
        // block 0's small parameters and first three weight fragments are requested before anything else of the item
        ParRegs<D> PR;
        float wa[D / 4], wb[D / 4], wc[D / 4];
        par_fetch<D>(PR, P.blk[0], tid);
        const float lastv = tid < 2 * D ? (tid < D ? P.last_w[tid] : P.last_b[tid - D]) : 0.f;
        FWREQ(wa, P.blk[0].in_w);
        FWREQ(wb, P.blk[0].in_w + D * D);
        FWREQ(wc, P.blk[0].in_w + 2 * D * D);
        enc_sync();
        ENC_MARK(g_fwd_marks, mk); ++mk;
        enc_decode<D>(PL, it, seq, tid, s_gid, s_first, s_pad);
        enc_sync();
        ENC_MARK(g_fwd_marks, mk); ++mk;
        if (tid < C::ROWS) {   // the sequence of a row and the item-local row its first token sits in (rows of a sequence are consecutive)
            const int gid = s_gid[tid], sid = gid >= 0 ? gid / S : -1;
            s_sid[tid] = sid;
            s_start[tid] = gid >= 0 ? tid - (gid - sid * S - s_first[tid]) : 0;
        }
        if (HEAD && tid < C::ROWS) {   // the rows' own item and their positive / negative (requested now, used after the last block)
            const int gid = s_gid[tid];
            int64_t it0 = 0, pr = 0, ng = 0;
            if (gid >= 0) { it0 = seq[gid]; pr = H.pos[gid] + H.e_off; ng = H.neg[gid] + H.e_off; }
            const bool real = it0 > 0 && it0 < H.R;
            const bool ok = real && pr > 0 && pr < H.R && ng > 0 && ng < H.R;
            s_item[tid] = real ? (int)it0 : 0;
            s_pos[tid] = ok ? (int)pr : 0;
            s_neg[tid] = ok ? (int)ng : 0;
        }
        // ---- x0 rows: from the tables (re_sasrec_embed fused in) or given
        if (em.E) {
            for (int f = tid; f < nrows * (D / 4); f += C::NT) {
                const int r = f / (D / 4), c4 = f % (D / 4);
                const int gid = s_gid[r];
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gid >= 0) {
                    const int64_t item = seq[gid];
                    if (item > 0 && item < em.R) {
                        const float4 v = reinterpret_cast<const float4*>(em.E + item * D)[c4];
                        const float4 p = reinterpret_cast<const float4*>(em.P + (int64_t)(gid % S) * D)[c4];
                        o.x = v.x * em.scale + p.x;
                        o.y = v.y * em.scale + p.y;
                        o.z = v.z * em.scale + p.z;
                        o.w = v.w * em.scale + p.w;
                        if (thresh) {
                            const uint32_t e = (uint32_t)((int64_t)gid * D + c4 * 4);
                            o.x = re_keep(seed, RE_STREAM_EMBED, e + 0, thresh) ? o.x * drop_scale : 0.f;
                            o.y = re_keep(seed, RE_STREAM_EMBED, e + 1, thresh) ? o.y * drop_scale : 0.f;
                            o.z = re_keep(seed, RE_STREAM_EMBED, e + 2, thresh) ? o.z * drop_scale : 0.f;
                            o.w = re_keep(seed, RE_STREAM_EMBED, e + 3, thresh) ? o.w * drop_scale : 0.f;
                        }
                    }
                }
                *reinterpret_cast<float4*>(bX + r * C::LS + 4 * c4) = o;
            }
        } else {
            TileRegs<D> R;
            tile_fetch_gid<D>(R, x0, s_gid, nrows, tid);
            tile_commit<D>(bX, R, nrows, tid);
        }
        par_commit<D>(s_par, PR, tid);
        if (tid < 2 * D) s_last[tid] = lastv;
        enc_sync();
        ENC_MARK(g_fwd_marks, mk); ++mk;

        TileRegs<D> HP, HN;
        for (int l = 0; l < L; ++l) {
            float* tp = TRAIN ? tape + (int64_t)l * T.per_block : nullptr;
            const float* par = s_par + (l & 1) * EP_NPAR * D;
            const bool more = l + 1 < L;
            const SasrecBlockParams Wn = P.blk[more ? l + 1 : l];   // the NEXT block's weights: requested two or more phases before use
            par_fetch<D>(PR, Wn, tid);                  // (unconditional: the last block re-reads its own -- a branch here hides the loads from the wait counting)
            TileRegs<D> TK0, TV0;
            if (npre && !split_hi) {   // this block's k, v of the prefix rows (tape; written by this workgroup a moment ago)
                tile_fetch<D>(TK0, tape + (int64_t)l * T.per_block + T.off_K + prow0 * D, 16 * npre, tid);
                tile_fetch<D>(TV0, tape + (int64_t)l * T.per_block + T.off_V + prow0 * D, 16 * npre, tid);
            }
            // ---- 1. Q-input = LN_a(x)
            if (r_e < nrows) {
                float mean, rstd;
                ln_row<D>(bX, bA, par + 0 * D, par + 1 * D, tid, mean, rstd);
                if (TRAIN && row_lead) {
                    float* st = tp + T.off_SA + (row0 + r_e) * 2;
                    st[0] = mean; st[1] = rstd;
                }
            }
            if (TRAIN) tile_store<D>(bX, tp + T.off_X + row0 * D, nrows, tid);
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 2. q, k, v projections (wa, wb, wc hold Wq, Wk, Wv); each register set is re-requested as soon as its product is done
            {
                const SasrecBlockParams W = P.blk[l];
                const float bq = par[2 * D + col], bk = par[3 * D + col], bv = par[4 * D + col];
                FWUSE(wa, W.in_w);
                gemm_rows<D>(bA, wa, lane, wr, it.nt, [&](int row, float v) { bQ[row * C::LS + col] = v + bq; });
                FWREQ(wa, W.out_w);                 // Wo
                FWUSE(wb, W.in_w + D * D);
                gemm_rows<D>(bX, wb, lane, wr, it.nt, [&](int row, float v) { bK[row * C::LS + col] = v + bk; });
                FWREQ(wb, W.w1);                    // W1
                FWUSE(wc, W.in_w + 2 * D * D);
                gemm_rows<D>(bX, wc, lane, wr, it.nt, [&](int row, float v) { bV[row * C::LS + col] = v + bv; });
                FWREQ(wc, W.w2);                    // W2
                if (TRAIN) tile_store<D>(bA, tp + T.off_A + row0 * D, nrows, tid);
                if (npre && !split_hi) {
                    tile_commit<D>(bK0, TK0, 16 * npre, tid);
                    tile_commit<D>(bV0, TV0, 16 * npre, tid);
                }
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            if (TRAIN && split_hi) {   // the first half's k, v of this block: published by its workgroup right after ITS projections
                if (tid == 0) enc_flag_wait(tflags, prow0 / 16, l, ferr);
                __syncthreads();
                tile_load_coh<D>(bK0, tp + T.off_K + prow0 * D, 32, tid);
                tile_load_coh<D>(bV0, tp + T.off_V + prow0 * D, 32, tid);
                enc_sync();
            }
            // ---- 3. scores = q k^T / sqrt(D) over the item's (row tile, key tile) pairs
            gemm_pairs<D>(bQ, bK, lane, wave, it, [&](int row, int key, float v) { sP[row * C::PLS + key] = v * inv_sqrt_d; }, bK0, npre);
            if (TRAIN) {
                tile_store<D>(bQ, tp + T.off_Q + row0 * D, nrows, tid);
                if (split_lo) {   // (device-scope stores: the second half's workgroup may sit on another XCD)
                    tile_store_coh<D>(bK, tp + T.off_K + row0 * D, nrows, tid);
                    tile_store_coh<D>(bV, tp + T.off_V + row0 * D, nrows, tid);
                } else {
                    tile_store<D>(bK, tp + T.off_K + row0 * D, nrows, tid);
                    tile_store<D>(bV, tp + T.off_V + row0 * D, nrows, tid);
                }
            }
            if (TRAIN && split_lo) {
                re_sync_full();    // (with vmcnt(0): the tiles have left this CU)
                if (tid == 0) enc_flag_set(tflags, row0 / 16, l);
            } else {
                enc_sync();
            }
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- softmax over the keys of the same sequence with j <= i (causal; explicit pad rows ARE keys), plus the virtual
            //      pad key in front of the sequence (multiplicity first, score q.b_k/sqrt(D), value b_v); dropout on the probabilities
            if (r_e < nrows) {
                const int i = r_e;
                const int gi = s_gid[i], n_out = s_first[i];
                // the row's keys are the columns [st, i] (own rows of the same sequence, consecutive in the item) plus, in a chained
                // part, the prefix columns in front (st = -16 npre there); key column jo is position n_out + jo - st of the sequence
                const int kpre = 16 * npre, st = s_start[i];
                const unsigned span = gi >= 0 ? (unsigned)(i - st) : 0u;
                float p[KPT];
                float mx = -INFINITY;
                unsigned okm = 0;
#pragma unroll
                for (int q = 0; q < KPT / 4; ++q) ld4(&p[4 * q], sP + i * C::PLS + j0_e + 4 * q);
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) {
                    const int jo = j0_e + jj - kpre;
                    const bool ok = gi >= 0 && (unsigned)(jo - st) <= span;
                    okm |= (ok ? 1u : 0u) << jj;
                    p[jj] = ok ? p[jj] : -INFINITY;
                    mx = fmaxf(mx, p[jj]);
                }
                float d = 0.f;
#pragma unroll
                for (int jj = 0; jj < C::CPT; ++jj) d = fmaf(bQ[i * C::LS + c0_e + jj], par[3 * D + c0_e + jj], d);
                d = row_sum<C::TPR>(d);
                const float spad = (gi >= 0 && n_out > 0) ? d * inv_sqrt_d : -INFINITY;
                mx = row_max<C::TPR>(fmaxf(mx, spad));
                float sum = 0.f;
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) {
                    p[jj] = ((okm >> jj) & 1u) ? expf(p[jj] - mx) : 0.f;
                    sum += p[jj];
                }
                sum = row_sum<C::TPR>(sum);
                const float epad = (spad == -INFINITY) ? 0.f : expf(spad - mx);
                sum += (float)n_out * epad;
                const float inv = (gi >= 0) ? 1.0f / sum : 0.f;
                const float ppad = epad * inv;
                float kept = (float)n_out;
                if (thresh) {   // each of the n_out pad keys has its own dropout bit (element (b, s_i, jj)); S <= 64: at most 64 / TPR per thread
                    int cnt = 0;
                    const uint32_t e0 = (uint32_t)((int64_t)gi * S);
#pragma unroll
                    for (int q = 0; q < 64 / C::TPR; ++q) {
                        const int jj = (tid % C::TPR) + q * C::TPR;
                        cnt += (jj < n_out && re_keep(seed, RE_STREAM_ATTN(l), e0 + (uint32_t)jj, thresh)) ? 1 : 0;
                    }
                    cnt = row_sum_i<C::TPR>(gi >= 0 ? cnt : 0);
                    kept = (float)cnt * drop_scale;
                }
                const float wv = (gi >= 0) ? ppad * kept : 0.f;
                if (row_lead) {
                    s_w[i] = wv;
                    if (TRAIN) {
                        float* pp = tp + T.off_PP + (row0 + i) * 2;
                        pp[0] = ppad; pp[1] = wv;
                    }
                }
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) p[jj] *= inv;
                if (TRAIN) {   // pre-dropout probabilities (0 outside the row's window)
                    float* dst = tp + T.off_P + (row0 + i) * EP_PW + j0_e;
#pragma unroll
                    for (int q = 0; q < KPT / 4; ++q)
                        reinterpret_cast<float4*>(dst)[q] = make_float4(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
                }
                if (thresh) {
                    const uint32_t e0 = (uint32_t)((int64_t)gi * S + n_out - st - kpre + j0_e);   // + jj: the key's position in the sequence
#pragma unroll
                    for (int jj = 0; jj < KPT; ++jj)
                        p[jj] = re_keep(seed, RE_STREAM_ATTN(l), e0 + (uint32_t)jj, thresh) ? p[jj] * drop_scale : 0.f;
                }
#pragma unroll
                for (int q = 0; q < KPT / 4; ++q)
                    *reinterpret_cast<float4*>(sP + i * C::PLS + j0_e + 4 * q) = make_float4(p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]);
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 4. o = A v + w * b_v
            {
                const float bv = par[4 * D + col];
                gemm_tx<D>(sP, bV, lane, wr, strip, it, [&](int row, float v) { bA[row * C::LS + col] = fmaf(s_w[row], bv, v); }, bV0, npre);
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 5. x1 = o Wo^T + bo + x
            {
                const float bo = par[5 * D + col];
                FWUSE(wa, P.blk[l].out_w);
                gemm_rows<D>(bA, wa, lane, wr, it.nt, [&](int row, float v) { bQ[row * C::LS + col] = v + bo + bX[row * C::LS + col]; });
                FWREQ(wa, Wn.in_w);                   // next block's Wq
                if (TRAIN) tile_store<D>(bA, tp + T.off_O + row0 * D, nrows, tid);
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 6. y = LN_f(x1)
            if (r_e < nrows) {
                float mean, rstd;
                ln_row<D>(bQ, bK, par + 6 * D, par + 7 * D, tid, mean, rstd);
                if (TRAIN && row_lead) {
                    float* st = tp + T.off_SF + (row0 + r_e) * 2;
                    st[0] = mean; st[1] = rstd;
                }
            }
            if (TRAIN) tile_store<D>(bQ, tp + T.off_X1 + row0 * D, nrows, tid);
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 7. hr = relu(dropout1(y W1^T + b1))
            {
                const float b1 = par[8 * D + col];
                FWUSE(wb, P.blk[l].w1);
                gemm_rows<D>(bK, wb, lane, wr, it.nt, [&](int row, float v) {
                    v += b1;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[row] * D + col);
                        v = re_keep(seed, RE_STREAM_FFN1(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    bV[row * C::LS + col] = fmaxf(v, 0.f);
                });
                FWREQ(wb, Wn.in_w + D * D);           // next block's Wk
                if (TRAIN) tile_store<D>(bK, tp + T.off_Y + row0 * D, nrows, tid);
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 8. x' = dropout2(hr W2^T + b2) + y, pad rows zeroed
            if (HEAD && !more) {   // the loss head's item rows E[pos], E[neg]: requested two phases before they are used
                tile_fetch_rows<D>(HP, H.E, s_pos, nrows, tid);
                tile_fetch_rows<D>(HN, H.E, s_neg, nrows, tid);
            }
            {
                const float b2 = par[9 * D + col];
                FWUSE(wc, P.blk[l].w2);
                gemm_rows<D>(bV, wc, lane, wr, it.nt, [&](int row, float v) {
                    v += b2;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[row] * D + col);
                        v = re_keep(seed, RE_STREAM_FFN2(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    v += bK[row * C::LS + col];
                    bX[row * C::LS + col] = s_pad[row] ? 0.f : v;
                });
                FWREQ(wc, Wn.in_w + 2 * D * D);       // next block's Wv
                if (TRAIN) tile_store<D>(bV, tp + T.off_HR + row0 * D, nrows, tid);
                if (more) par_commit<D>(s_par + ((l + 1) & 1) * EP_NPAR * D, PR, tid);   // (the other half: this block's readers use `par`)
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
        }
        // ---- u = LN_last(x_L)
        if (r_e < nrows) {
            float mean, rstd;
            ln_row<D>(bX, bA, s_last, s_last + D, tid, mean, rstd);
            if (TRAIN && row_lead) {
                float* st = tape + T.off_SL + (row0 + r_e) * 2;
                st[0] = mean; st[1] = rstd;
                if (HO) { HO->mean[r_e] = mean; HO->rstd[r_e] = rstd; }
            }
        }
        if (TRAIN) tile_store<D>(bX, tape + T.off_XL + row0 * D, nrows, tid);
        if (HEAD) {   // (bQ, bK are dead after the last block)
            tile_commit<D>(bQ, HP, nrows, tid);
            tile_commit<D>(bK, HN, nrows, tid);
        }
        enc_sync();
        ENC_MARK(g_fwd_marks, mk); ++mk;
        tile_store_gid<D>(bA, u, s_gid, nrows, tid);
        if (HEAD && r_e < nrows) {
            // ---- loss head: pl = <u, E[pos]>, nl = <u, E[neg]>; the row's three gradient rows and their destination keys
            const int i = r_e;
            const int pr = s_pos[i], ng = s_neg[i];
            const bool ok = pr != 0;
            float a[C::CPT], b[C::CPT], d[C::CPT];
#pragma unroll
            for (int q = 0; q < C::CPT / 4; ++q) {
                ld4(&a[4 * q], bA + i * C::LS + c0_e + 4 * q);
                ld4(&b[4 * q], bQ + i * C::LS + c0_e + 4 * q);
                ld4(&d[4 * q], bK + i * C::LS + c0_e + 4 * q);
            }
            float pl = 0.f, nl = 0.f;
#pragma unroll
            for (int q = 0; q < C::CPT; ++q) { pl = fmaf(a[q], b[q], pl); nl = fmaf(a[q], d[q], nl); }
            pl = row_sum<C::TPR>(pl);
            nl = row_sum<C::TPR>(nl);
            const float gs = 1.0f / (float)H.count[0];
            float dpl, dnl;
            if (H.kind == RE_LOSS_BCE) { dpl = -re_sigmoid(-pl) * gs; dnl = re_sigmoid(nl) * gs; }
            else { const float sg = re_sigmoid(nl - pl) * gs; dpl = -sg; dnl = sg; }
            if (!ok) { dpl = 0.f; dnl = 0.f; }
            const int64_t NRH = 16 * enc_plan_max_tiles(B, S), r = row0 + i;
            float* du = H.dU_rows + r * D + c0_e;
            float* gp = H.g_rows + (NRH + r) * D + c0_e;
            float* gn = H.g_rows + (2 * NRH + r) * D + c0_e;
#pragma unroll
            for (int q = 0; q < C::CPT / 4; ++q) {
                const float4 dv = make_float4(fmaf(dpl, b[4 * q], dnl * d[4 * q]), fmaf(dpl, b[4 * q + 1], dnl * d[4 * q + 1]),
                                              fmaf(dpl, b[4 * q + 2], dnl * d[4 * q + 2]), fmaf(dpl, b[4 * q + 3], dnl * d[4 * q + 3]));
                reinterpret_cast<float4*>(du)[q] = dv;
                if (HO) *reinterpret_cast<float4*>(HO->du + i * C::LS + c0_e + 4 * q) = dv;
                if (ok) {
                    reinterpret_cast<float4*>(gp)[q] = make_float4(dpl * a[4 * q], dpl * a[4 * q + 1], dpl * a[4 * q + 2], dpl * a[4 * q + 3]);
                    reinterpret_cast<float4*>(gn)[q] = make_float4(dnl * a[4 * q], dnl * a[4 * q + 1], dnl * a[4 * q + 2], dnl * a[4 * q + 3]);
                }
            }
            if (row_lead) {
                H.keys[r] = s_item[i];
                H.keys[NRH + r] = pr;
                H.keys[2 * NRH + r] = ng;
                if (ok) head_sum += (H.kind == RE_LOSS_BCE) ? re_softplus(-pl) + re_softplus(nl) : re_softplus(nl - pl);
            }
        }
        if (fill_pads) {
            // positions in front of a sequence's first row are pads: u = LN_last(0) = beta_last (what the reference's encode returns there)
            for (int r = 0; r < nrows; ++r) {
                const int gid = s_gid[r], first = s_first[r];
                if (gid < 0 || first == 0 || gid - s_sid[r] * S != first) continue;   // (workgroup-uniform)
                for (int f = tid; f < first * (D / 4); f += C::NT)
                    reinterpret_cast<float4*>(u + (int64_t)(gid - first) * D)[f] = reinterpret_cast<const float4*>(s_last + D)[f % (D / 4)];
            }
        }
        }   // chained parts
        if (HEAD) {
            // The item's loss goes into a 64-bit fixed-point word (2^-30 units; integer adds commute: the total does not depend on the
            // order the items arrive in), then the item takes a ticket in the next word: low half = arrivals, high half = how many
            // partials were NaN / Inf / beyond the fixed-point range -- a diverged model.  Such a partial is NOT converted (llrint of a
            // NaN is 0 or garbage): the finishing item then writes NaN, as torch's mean would.  Both are device-scope atomics ordered by a
            // data dependency (the ticket's operand is made from the add's return value); no fence: a release fence here would write
            // back the XCD's whole L2 once per item.  The finishing item leaves both words zero for the next launch.
            float t = re_wave_sum(head_sum);
            if (lane == 0) s_red[wave] = t;
            __syncthreads();
            if (tid == 0) {
                double part = 0.0;
#pragma unroll
                for (int w = 0; w < C::NW; ++w) part += (double)s_red[w];
                const bool finite = part == part && fabs(part) < 4294967296.0;
                const unsigned long long add = finite ? (unsigned long long)(long long)llrint(part * 1073741824.0) : 0ull;
                const unsigned long long old = __hip_atomic_fetch_add(H.acc, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long one = 1ull + (finite ? 0ull : (1ull << 32)) + (old & 0ull);
                const unsigned long long ticket = __hip_atomic_fetch_add(H.acc + 1, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(ticket & 0xFFFFFFFFull) == PL.hdr[0] - 1) {
                    const unsigned long long tot = __hip_atomic_exchange(H.acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool bad = ((ticket + one) >> 32) != 0ull;
                    const int cnt = H.count[0];
                    H.loss[0] = (cnt > 0 && !bad) ? (float)((double)(long long)tot * (1.0 / 1073741824.0) / (double)cnt) : __builtin_nanf("");   // mean over an empty set is NaN, as torch's
                    __hip_atomic_store(H.acc + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
                }
            }
        }
    }
}
