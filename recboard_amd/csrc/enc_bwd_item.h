// K6/K7 backward, the work of ONE item (device function; kernels: enc_bwd.hip, enc_step.hip): all blocks (l = L-1 .. 0), one work item per workgroup iteration, everything in LDS.
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
#pragma once
#include <math.h>

#include "enc_common.h"

template <int D>
__device__ __forceinline__ float colsum(const float* tile, int tid, int nrows) {
    using C = EC<D>;
    const int c = tid % D, r0 = (tid / D) * C::RPW;
    float s = 0.f;
    if (r0 < nrows) {
#pragma unroll
        for (int i = 0; i < C::RPW; ++i) s += tile[(r0 + i) * C::LS + c];
    }
    return s;
}
// sum over the thread's rows of w[row] * tile[row][col]
template <int D>
__device__ __forceinline__ float colsum_w(const float* tile, const float* w, int tid, int nrows) {
    using C = EC<D>;
    const int c = tid % D, r0 = (tid / D) * C::RPW;
    float s = 0.f;
    if (r0 < nrows) {
#pragma unroll
        for (int i = 0; i < C::RPW; ++i) s = fmaf(w[r0 + i], tile[(r0 + i) * C::LS + c], s);
    }
    return s;
}
// sum over the thread's rows of dy * xhat, xhat = (x - mean[row]) * rstd[row]
template <int D>
__device__ __forceinline__ float colsum_xhat(const float* dy, const float* x, const float* mean, const float* rstd, int tid, int nrows) {
    using C = EC<D>;
    const int c = tid % D, r0 = (tid / D) * C::RPW;
    float s = 0.f;
    if (r0 < nrows) {
#pragma unroll
        for (int i = 0; i < C::RPW; ++i) {
            const int r = r0 + i;
            s = fmaf(dy[r * C::LS + c], (x[r * C::LS + c] - mean[r]) * rstd[r], s);
        }
    }
    return s;
}
// LayerNorm backward for this thread's row slice: dst (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma
template <int D, bool ACCUM>
__device__ __forceinline__ void ln_bwd_row(const float* dy, const float* x, float* dst, const float* __restrict__ gamma, const float* mean,
                                           const float* rstd, int tid) {
    using C = EC<D>;
    const int r = tid / C::TPR, c0 = (tid % C::TPR) * C::CPT;
    float d[C::CPT], xv[C::CPT];
#pragma unroll
    for (int q = 0; q < C::CPT / 4; ++q) {
        ld4(&d[4 * q], dy + r * C::LS + c0 + 4 * q);
        ld4(&xv[4 * q], x + r * C::LS + c0 + 4 * q);
    }
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < C::CPT; ++i) {
        xv[i] = (xv[i] - mu) * rs;
        d[i] *= gamma[c0 + i];
        s1 += d[i];
        s2 = fmaf(d[i], xv[i], s2);
    }
    s1 = row_sum<C::TPR>(s1) * (1.0f / D);
    s2 = row_sum<C::TPR>(s2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < C::CPT; ++i) {
        const float v = rs * (d[i] - s1 - xv[i] * s2);
        if (ACCUM) dst[r * C::LS + c0 + i] += v; else dst[r * C::LS + c0 + i] = v;
    }
}

__device__ __forceinline__ void stats_fetch(float2& r, const float* st, int nrows, int tid) {
    tid = enc_opaque(tid);
    gcf_t sp = g_launder(st);
    r = make_float2(0.f, 0.f);
    if (tid < nrows) { r.x = sp[2 * tid]; r.y = sp[2 * tid + 1]; }
}

// (see enc_fwd.hip: D = 64 keeps three fragment sets in flight, D = 128 loads a fragment where it is used)
#define BWREQ(reg, ptr) do { if (D == 64) wfrag_n<D>(reg, ptr, strip, lane); } while (0)
#define BWUSE(reg, ptr) do { if (D != 64) wfrag_n<D>(reg, ptr, strip, lane); } while (0)

#ifndef BWD_UNCOND
#define BWD_UNCOND 0   // (measured: 46.0 vs 48.2 us -- the backward already orders its requests a phase ahead of their use)
#endif

template <int D>
__device__ __forceinline__ void enc_bwd_item(const float* __restrict__ dIn, const int64_t* __restrict__ seq, int B, int S, int L,
                                             const SasrecParams& P, float drop_scale, uint32_t thresh, uint32_t seed,
                                             const float* __restrict__ tape, const EncTape& T, const EncPlan& PL, float* __restrict__ dOut,
                                             float* __restrict__ gtape, float* __restrict__ slab, int fuse_embed, float emb_scale, int in_rows,
                                             float* __restrict__ dOutRows, float* lds, int wi, int k, const EncHandoff* HO = nullptr) {
    // in_rows: dIn is indexed by the plan's compact rows instead of (b, s).  dOutRows (optional): the output rows once more, in compact order.
    using C = EC<D>;
    constexpr int KPT = C::KPT;
    float* b0 = lds;
    float* b1 = b0 + C::BUF;
    float* b2 = b1 + C::BUF;
    float* b3 = b2 + C::BUF;
    float* b4 = b3 + C::BUF;
    float* sP = b4 + C::BUF;
    float* sD = sP + C::PBUF;
    float* bK0 = sD + C::PBUF;                 // prefix k / v tiles (32 rows) of a chained part / the second half of a split sequence
    float* bV0 = bK0 + C::PRE;
    __shared__ float s_mean[C::ROWS], s_rstd[C::ROWS];
    __shared__ float s_ppad[C::ROWS], s_w[C::ROWS], s_cpad[C::ROWS];   // virtual pad key: prob of one copy, total kept weight, dS
    __shared__ int o_gid[C::ROWS], o_first[C::ROWS], o_pad[C::ROWS], o_sid[C::ROWS], o_start[C::ROWS];
    int *s_gid = HO ? HO->gid : o_gid, *s_first = HO ? HO->first : o_first, *s_pad = HO ? HO->pad : o_pad, *s_sid = HO ? HO->sid : o_sid,
        *s_start = HO ? HO->start : o_start;   // (left by the forward of the same launch: enc_step_k)
    __shared__ float s_par[2 * EP_NPAR * D], s_last[D];

    const int tid0 = threadIdx.x;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (a scalar: the wave's tile choices are scalar branches)
        const int strip = wave % C::NS, wr = wave / C::NS;
        const int c = lane & 15, col = 16 * strip + c;
        const int r_e = tid / C::TPR, c0_e = (tid % C::TPR) * C::CPT, j0_e = (tid % C::TPR) * KPT;
        const bool row_lead = (tid % C::TPR) == 0;
        const EncItem whole = enc_item(PL, wi);
        // chained parts of a sequence with more rows than the LDS holds (D = 128; enc_fwd.hip), LAST part first: the later rows'
        // queries also attend to the earlier rows (prefix key tiles k, v from the tape), and what they contribute to those rows' dK,
        // dV is left on the gradient tape for the earlier part to add.
        const int nsub = C::MAXT < 4 ? (whole.nt + C::MAXT - 1) / C::MAXT : 1;
        const bool split_lo = whole.kind == 2, split_hi = whole.kind == 3;   // halves of a sequence split over two workgroups (enc_common.h)
        float* tflags = const_cast<float*>(tape) + T.off_FLAGS;
        const int64_t ferr = enc_plan_max_tiles(B, S) * EP_FLAG_WORDS;
        for (int hs = nsub - 1; hs >= 0; --hs) {
        const EncItem it = EncItem{whole.tile0 + hs * C::MAXT, whole.nt - hs * C::MAXT < C::MAXT ? whole.nt - hs * C::MAXT : C::MAXT, whole.kind};
        const int npre = split_hi ? 2 : (C::MAXT < 4 ? hs * C::MAXT : 0);     // prefix key tiles
        const bool has_succ = split_lo || (C::MAXT < 4 && hs + 1 < nsub);    // a later part leaves partial dK / dV for these rows
        const int64_t prow0 = (int64_t)(whole.tile0 - (split_hi ? 2 : 0)) * 16;   // compact row of the sequence's first row
        if (hs + 1 < nsub) re_sync_full();                   // (a full barrier: the later part's gradient-tape stores have completed)
        const bool first_part = k == 0 && hs == nsub - 1;    // the workgroup's first flush of its vector-gradient slab
        const int nt = it.nt, nrows = 16 * nt;
        const int64_t row0 = (int64_t)it.tile0 * 16;
        int mk = 0; (void)mk;
        // the last block's small parameters and first three weight fragments are requested before anything else of the item
        ParRegs<D> PR;
        float wa[D / 4], wb[D / 4], wc[D / 4];
        par_fetch<D>(PR, P.blk[L - 1], tid);
        const float lastv = tid < D ? P.last_w[tid] : 0.f;
        BWREQ(wa, P.blk[L - 1].w2);
        BWREQ(wb, P.blk[L - 1].w1);
        BWREQ(wc, P.blk[L - 1].out_w);
        // handed over by the forward of this launch (the item it has just finished: the last part of a chained item): metadata,
        // lastLN's statistics, x_L (still in the first tile buffer) and the upstream gradient rows -- nothing of the prologue is read back
        const bool handed = HO != nullptr && hs == nsub - 1;
        TileRegs<D> T0, T1;
        float2 ST;
        if (handed) tile_fetch<D>(T1, tape + (int64_t)(L - 1) * T.per_block + T.off_HR + row0 * D, nrows, tid);
        enc_sync();
        ENC_MARK(g_bwd_marks, mk); ++mk;
        if (!handed) enc_decode<D>(PL, it, seq, tid, s_gid, s_first, s_pad);
        enc_sync();
        ENC_MARK(g_bwd_marks, mk); ++mk;
        if (!handed && tid < C::ROWS) {   // the sequence of a row and the item-local row its first token sits in (as in the forward)
            const int gid = s_gid[tid], sid = gid >= 0 ? gid / S : -1;
            s_sid[tid] = sid;
            s_start[tid] = gid >= 0 ? tid - (gid - sid * S - s_first[tid]) : 0;
        }
        par_commit<D>(s_par + ((L - 1) & 1) * EP_NPAR * D, PR, tid);
        if (tid < D) s_last[tid] = lastv;
        if (handed) {
            if (r_e < nrows) {   // x_L: first tile buffer -> second; dU: hand-off tile -> first (each thread its own row slice)
#pragma unroll
                for (int q = 0; q < C::CPT / 4; ++q) {
                    float t[4];
                    ld4(t, b0 + r_e * C::LS + c0_e + 4 * q);
                    *reinterpret_cast<float4*>(b1 + r_e * C::LS + c0_e + 4 * q) = make_float4(t[0], t[1], t[2], t[3]);
                    ld4(t, HO->du + r_e * C::LS + c0_e + 4 * q);
                    *reinterpret_cast<float4*>(b0 + r_e * C::LS + c0_e + 4 * q) = make_float4(t[0], t[1], t[2], t[3]);
                }
            }
            if (tid < C::ROWS) { s_mean[tid] = HO->mean[tid]; s_rstd[tid] = HO->rstd[tid]; }
        } else {
            if (in_rows) tile_fetch<D>(T0, dIn + row0 * D, nrows, tid);
            else tile_fetch_gid<D>(T0, dIn, s_gid, nrows, tid);
            tile_fetch<D>(T1, tape + T.off_XL + row0 * D, nrows, tid);
            stats_fetch(ST, tape + T.off_SL + row0 * 2, nrows, tid);
            tile_commit<D>(b0, T0, nrows, tid);
            tile_commit<D>(b1, T1, nrows, tid);
            if (tid < C::ROWS) { s_mean[tid] = ST.x; s_rstd[tid] = ST.y; }
            tile_fetch<D>(T1, tape + (int64_t)(L - 1) * T.per_block + T.off_HR + row0 * D, nrows, tid);
        }
        enc_sync();
        ENC_MARK(g_bwd_marks, mk); ++mk;
        float accV[EG_NVEC];
#pragma unroll
        for (int v = 0; v < EG_NVEC; ++v) accV[v] = 0.f;
        // ---- u = LN_last(x_L): dgamma / dbeta, then dx_L in place
        accV[10] = colsum_xhat<D>(b0, b1, s_mean, s_rstd, tid, nrows);
        accV[11] = colsum<D>(b0, tid, nrows);
        enc_sync();
        ENC_MARK(g_bwd_marks, mk); ++mk;
        if (r_e < nrows) ln_bwd_row<D, false>(b0, b1, b0, s_last, s_mean, s_rstd, tid);
        enc_sync();
        ENC_MARK(g_bwd_marks, mk); ++mk;

        // Order inside a phase: commit what was requested earlier -> products -> request what later phases need -> stores.
        // (The memory counter retires in order: a wait for a request also waits for everything issued before it, so stores go last
        // and every request is at least one phase older than its first use.)
        for (int l = L - 1; l >= 0; --l) {
            const SasrecBlockParams W = P.blk[l];
            const SasrecBlockParams Wn = P.blk[l > 0 ? l - 1 : 0];   // the block after this one
            const bool more = l > 0;
            const float* par = s_par + (l & 1) * EP_NPAR * D;
            const float* tp = tape + (int64_t)l * T.per_block;
            float* gp = gtape + (int64_t)l * EG_NMAT * NR * D + row0 * D;
            if (l != L - 1) {
                accV[10] = 0.f; accV[11] = 0.f;
            }
            if (BWD_UNCOND || more) par_fetch<D>(PR, Wn, tid);   // (unconditional, like every request below: behind a branch the wait counting cannot see them and drains the queue)
            // ---- pad mask of the block output (x'[pad] = 0) and dO2 = dX' * dropout2 mask      [T1 = HR in flight]
            if (r_e < nrows) {
                const bool dead = s_pad[r_e] != 0;
#pragma unroll
                for (int i = 0; i < C::CPT; ++i) {
                    float v = dead ? 0.f : b0[r_e * C::LS + c0_e + i];
                    b0[r_e * C::LS + c0_e + i] = v;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[r_e] * D + c0_e + i);
                        v = re_keep(seed, RE_STREAM_FFN2(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    b2[r_e * C::LS + c0_e + i] = v;
                }
            }
            tile_commit<D>(b1, T1, nrows, tid);   // HR
            tile_fetch<D>(T0, tp + T.off_X1 + row0 * D, nrows, tid);
            stats_fetch(ST, tp + T.off_SF + row0 * 2, nrows, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- A. FFN second map: db2; dH = (dO2 W2) * (hr > 0) * scale        (wa = W2)
            BWUSE(wa, W.w2);
            gemm_rows<D>(b2, wa, lane, wr, nt, [&](int row, float v) {
                b3[row * C::LS + col] = (b1[row * C::LS + col] > 0.f) ? v * drop_scale : 0.f;
            });
            BWREQ(wa, W.in_w);                  // Wq
            accV[5] = colsum<D>(b2, tid, nrows);
            tile_store<D>(b2, gp + 0 * NR * D, nrows, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- B. FFN first map: db1; dY = dH W1 + dX'        (wb = W1)
            tile_commit<D>(b1, T0, nrows, tid);   // X1  (HR's last readers are behind the barrier above)
            if (tid < C::ROWS) { s_mean[tid] = ST.x; s_rstd[tid] = ST.y; }
            BWUSE(wb, W.w1);
            gemm_rows<D>(b3, wb, lane, wr, nt, [&](int row, float v) { b0[row * C::LS + col] += v; });
            BWREQ(wb, W.in_w + D * D);          // Wk
            tile_fetch<D>(T0, tp + T.off_V + row0 * D, nrows, tid);
            stats_fetch(ST, tp + T.off_PP + row0 * 2, nrows, tid);
            float pq[KPT];   // this thread's slice of the saved probabilities
            {
#pragma unroll
                for (int q = 0; q < KPT; ++q) pq[q] = 0.f;
                if (r_e < nrows) {
                    const float* src = tp + T.off_P + (row0 + enc_opaque(r_e)) * EP_PW + enc_opaque(j0_e);
                    if (KPT % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < KPT / 4; ++q) ld4(&pq[4 * q], src + 4 * q);
                    } else {
#pragma unroll
                        for (int q = 0; q < KPT; ++q) pq[q] = src[q];
                    }
                }
            }
            accV[4] = colsum<D>(b3, tid, nrows);
            tile_store<D>(b3, gp + 1 * NR * D, nrows, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- C. LN_f backward: dgamma_f, dbeta_f, dX1 (in place in b0)
            accV[8] = colsum_xhat<D>(b0, b1, s_mean, s_rstd, tid, nrows);
            accV[9] = colsum<D>(b0, tid, nrows);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            if (r_e < nrows) ln_bwd_row<D, false>(b0, b1, b0, par + 6 * D, s_mean, s_rstd, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- D. out_proj: dbo; dO = dX1 Wo        (wc = Wo)
            BWUSE(wc, W.out_w);
            gemm_rows<D>(b0, wc, lane, wr, nt, [&](int row, float v) { b3[row * C::LS + col] = v; });
            BWREQ(wc, W.in_w + 2 * D * D);      // Wv
            // ---- E. attention: V, P; Pd = P * mask
            tile_commit<D>(b1, T0, nrows, tid);   // V  (X1's last readers, phase C, are behind the barrier above)
            if (npre) {                           // the prefix rows' k, v of this block (tape; written by the forward)
                if (split_hi) {                   // (by the other half's workgroup, with device-scope stores: read the same way)
                    tile_load_coh<D>(bV0, tp + T.off_V + prow0 * D, 16 * npre, tid);
                    tile_load_coh<D>(bK0, tp + T.off_K + prow0 * D, 16 * npre, tid);
                } else {
                    TileRegs<D> TP;
                    tile_fetch<D>(TP, tp + T.off_V + prow0 * D, 16 * npre, tid);
                    tile_commit<D>(bV0, TP, 16 * npre, tid);
                    tile_fetch<D>(TP, tp + T.off_K + prow0 * D, 16 * npre, tid);
                    tile_commit<D>(bK0, TP, 16 * npre, tid);
                }
            }
            if (tid < C::ROWS) { s_ppad[tid] = ST.x; s_w[tid] = ST.y; }
            tile_fetch<D>(T0, tp + T.off_K + row0 * D, nrows, tid);
            tile_fetch<D>(T1, tp + T.off_Q + row0 * D, nrows, tid);
            if (r_e < nrows) {
                const int i = r_e;
                const int gi = s_gid[i];
                // key column j is position first + j - start - 16 npre of the sequence (enc_fwd_item.h); the probability is 0 outside the
                // row's window (the forward stored zeros there), so the mask is regenerated for every column without a branch
                const uint32_t e0 = (uint32_t)((int64_t)gi * S + s_first[i] - s_start[i] - 16 * npre + j0_e);
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) {
                    const int j = j0_e + jj;
                    const float p = pq[jj];
                    const float m = !thresh ? 1.0f : re_keep(seed, RE_STREAM_ATTN(l), e0 + (uint32_t)jj, thresh) ? drop_scale : 0.f;
                    sP[i * C::PLS + j] = p;
                    sD[i * C::PLS + j] = p * m;
                }
            }
            accV[3] = colsum<D>(b0, tid, nrows);
            tile_store<D>(b0, gp + 2 * NR * D, nrows, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // dV = Pd^T dO; d b_v through the virtual pad key: sum_i w_i dO_i
            gemm_ttx<D>(sD, b3, lane, wr, strip, it, [&](int row, float v) { b2[row * C::LS + col] = v; }, npre);
            if (npre)                             // what these rows' queries add to the PREFIX rows' dV (-> b4, free until phase F)
                gemm_ttx_pre<D>(sD, b3, lane, wr, strip, it, npre, [&](int row, float v) { b4[row * C::LS + col] = v; });
            accV[2] = colsum_w<D>(b3, s_w, tid, nrows);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // dP = (dO V^T) * mask factor Pd / P (0, or 1 / (1 - p)), over the item's tile pairs (P is 0 elsewhere)
            gemm_pairs<D>(b3, b1, lane, wave, it, [&](int row, int key, float v) {
                const float p = sP[row * C::PLS + key];
                const float pd = sD[row * C::PLS + key];
                sD[row * C::PLS + key] = (p != 0.f) ? v * (pd / p) : 0.f;
            }, bV0, npre);
            float* gpre = gtape + (int64_t)l * EG_NMAT * NR * D + prow0 * D;   // the prefix rows of the gradient tape
            if (npre) {                                                                             // partial dV of the prefix rows
                if (split_hi) tile_store_coh<D>(b4, gpre + 5 * NR * D, 16 * npre, tid);
                else tile_store<D>(b4, gpre + 5 * NR * D, 16 * npre, tid);
            }
            if (has_succ && !split_lo) tile_add_global<D>(b2, gp + 5 * NR * D, nrows, tid);         // + what the later part left for these rows
            enc_sync();
            if (!split_lo) {   // (a split sequence's first half adds the other workgroup's partials after phase F, under the flag)
                accV[2] += colsum<D>(b2, tid, nrows);
                tile_store<D>(b2, gp + 5 * NR * D, nrows, tid);
            }
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // dS = P (dP - rowsum(dP P)) / sqrt(D), the virtual pad key included in the row sum
            if (r_e < nrows) {
                const int i = r_e;
                float dp[KPT], pp[KPT];
                float s = 0.f;
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) {
                    dp[jj] = sD[i * C::PLS + j0_e + jj];
                    pp[jj] = sP[i * C::PLS + j0_e + jj];
                    if (pp[jj] == 0.f) dp[jj] = 0.f;
                    s = fmaf(dp[jj], pp[jj], s);
                }
                s = row_sum<C::TPR>(s);
                // virtual pad key: upstream grad of each copy = (dO_i . b_v) * mask; t = dO_i . b_v
                float t = 0.f;
#pragma unroll
                for (int jj = 0; jj < C::CPT; ++jj) t = fmaf(b3[i * C::LS + c0_e + jj], par[4 * D + c0_e + jj], t);
                t = row_sum<C::TPR>(t);
                const float wv = s_w[i], ppad = s_ppad[i];
                s = fmaf(t, wv, s);                                            // rowdot includes the pad copies
                if (row_lead) s_cpad[i] = (wv * t - (float)s_first[i] * ppad * s) * inv_sqrt_d;   // sum of dS over the copies
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) sD[i * C::PLS + j0_e + jj] = pp[jj] * (dp[jj] - s) * inv_sqrt_d;
            }
            enc_sync();   // dS complete; dO (b3), V (b1) and P no longer needed
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- F. dQ = dS K (+ dS_pad b_k) -> b4 ; dK = dS^T Q -> b1
            tile_commit<D>(b1, T0, nrows, tid);   // K
            tile_commit<D>(b3, T1, nrows, tid);   // Q
            tile_fetch<D>(T0, tp + T.off_X + row0 * D, nrows, tid);
            stats_fetch(ST, tp + T.off_SA + row0 * 2, nrows, tid);
            if (BWD_UNCOND || more) tile_fetch<D>(T1, tape + (int64_t)(more ? l - 1 : 0) * T.per_block + T.off_HR + row0 * D, nrows, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            {
                const float bkc = par[3 * D + col];
                gemm_tx<D>(sD, b1, lane, wr, strip, it, [&](int row, float v) { b4[row * C::LS + col] = fmaf(s_cpad[row], bkc, v); }, bK0, npre);
            }
            accV[1] = colsum_w<D>(b3, s_cpad, tid, nrows);   // d b_k through the virtual pad key: sum_i dS_pad_i q_i
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            gemm_ttx<D>(sD, b3, lane, wr, strip, it, [&](int row, float v) { b1[row * C::LS + col] = v; }, npre);
            if (npre)                             // ... and to the prefix rows' dK (-> bV0: the prefix v is no longer needed)
                gemm_ttx_pre<D>(sD, b3, lane, wr, strip, it, npre, [&](int row, float v) { bV0[row * C::LS + col] = v; });
            accV[0] = colsum<D>(b4, tid, nrows);
            tile_store<D>(b4, gp + 3 * NR * D, nrows, tid);
            enc_sync();
            if (npre) {                                                                             // partial dK of the prefix rows
                if (split_hi) {
                    tile_store_coh<D>(bV0, gpre + 4 * NR * D, 16 * npre, tid);
                    re_sync_full();    // (with vmcnt(0): both partial tiles of this block have left this CU)
                    if (tid == 0) enc_flag_set(tflags, prow0 / 16, 4 + l);
                } else {
                    tile_store<D>(bV0, gpre + 4 * NR * D, 16 * npre, tid);
                }
            }
            if (split_lo) {   // the second half's partial dK, dV for these rows: published by its workgroup at the end of ITS phase F
                if (tid == 0) enc_flag_wait(tflags, row0 / 16, 4 + l, ferr);
                __syncthreads();
                tile_add_coh<D>(b1, gp + 4 * NR * D, nrows, tid);
                tile_add_coh<D>(b2, gp + 5 * NR * D, nrows, tid);
                enc_sync();
                accV[2] += colsum<D>(b2, tid, nrows);
                tile_store<D>(b2, gp + 5 * NR * D, nrows, tid);
            } else if (has_succ) {
                tile_add_global<D>(b1, gp + 4 * NR * D, nrows, tid);
                enc_sync();
            }
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- G. projections: dbq/dbk; dA1 = dQ Wq -> b3; dX (b0) += dK Wk + dV Wv        (wa, wb, wc = Wq, Wk, Wv)
            BWUSE(wa, W.in_w);
            gemm_rows<D>(b4, wa, lane, wr, nt, [&](int row, float v) { b3[row * C::LS + col] = v; });
            if (BWD_UNCOND || more) BWREQ(wa, Wn.w2);
            BWUSE(wb, W.in_w + D * D);
            gemm_rows<D>(b1, wb, lane, wr, nt, [&](int row, float v) { b0[row * C::LS + col] += v; });
            if (BWD_UNCOND || more) BWREQ(wb, Wn.w1);
            BWUSE(wc, W.in_w + 2 * D * D);
            gemm_rows<D>(b2, wc, lane, wr, nt, [&](int row, float v) { b0[row * C::LS + col] += v; });
            if (BWD_UNCOND || more) BWREQ(wc, Wn.out_w);
            accV[1] += colsum<D>(b1, tid, nrows);
            tile_store<D>(b1, gp + 4 * NR * D, nrows, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- H. LN_a backward: dgamma_a, dbeta_a; dX += LN_a'(dA1)
            tile_commit<D>(b4, T0, nrows, tid);   // X
            if (tid < C::ROWS) { s_mean[tid] = ST.x; s_rstd[tid] = ST.y; }
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            accV[6] = colsum_xhat<D>(b3, b4, s_mean, s_rstd, tid, nrows);
            accV[7] = colsum<D>(b3, tid, nrows);
            if (r_e < nrows) ln_bwd_row<D, true>(b3, b4, b0, par + 0 * D, s_mean, s_rstd, tid);
            if (more) par_commit<D>(s_par + ((l - 1) & 1) * EP_NPAR * D, PR, tid);
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            // ---- this block's vector gradients: row-group partials -> one value per column (fixed order) -> the workgroup's slab
            {
                float* red = b1;   // [EG_NVEC][CG][D]  (12 * 512 floats; b1 and b2 are free and adjacent)
#pragma unroll
                for (int v = 0; v < EG_NVEC; ++v) red[v * C::NT + tid] = accV[v];
                enc_sync();
                ENC_MARK(g_bwd_marks, mk); ++mk;
                float* sl = slab + ((int64_t)blockIdx.x * L + l) * EG_NVEC * D;
                for (int e = tid; e < EG_NVEC * D; e += C::NT) {
                    const int v = e / D, cc = e % D;
                    float s = red[v * C::NT + cc];
#pragma unroll
                    for (int i = 1; i < C::CG; ++i) s += red[v * C::NT + i * D + cc];
                    sl[e] = first_part ? s : sl[e] + s;
                }
                enc_sync();
                ENC_MARK(g_bwd_marks, mk); ++mk;
            }
        }
        if (fuse_embed) {
            // re_sasrec_embed_bwd fused in: pad rows -> 0, the embedding's dropout mask; the rows go out scaled by sqrt(D) as
            // item-gradient contributions (the position-table gradient is summed from them by enc_wgrad.hip)
            if (r_e < nrows) {
                const bool dead = s_pad[r_e] != 0;
#pragma unroll
                for (int i = 0; i < C::CPT; ++i) {
                    float v = dead ? 0.f : b0[r_e * C::LS + c0_e + i];
                    if (thresh && !dead) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[r_e] * D + c0_e + i);
                        v = re_keep(seed, RE_STREAM_EMBED, e, thresh) ? v * drop_scale : 0.f;
                    }
                    b0[r_e * C::LS + c0_e + i] = v;
                }
            }
            enc_sync();
            ENC_MARK(g_bwd_marks, mk); ++mk;
            tile_store_gid<D>(b0, dOut, s_gid, nrows, tid, emb_scale);
            if (dOutRows) tile_store<D>(b0, dOutRows + row0 * D, nrows, tid, emb_scale);
        } else {
            tile_store_gid<D>(b0, dOut, s_gid, nrows, tid);
            if (dOutRows) tile_store<D>(b0, dOutRows + row0 * D, nrows, tid);
        }
        }   // chained parts
    }
}
