// K6/K7 forward: the whole SASRec encoder (embedding front end, all blocks, lastLN) for one work item per workgroup,
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

// Weight fragments.  D = 64: three register sets, each re-requested as soon as its product is done -- two or more phases before
// its next use (WREQ; WUSE is empty).  D = 128: a fragment is 32 registers and three sets in flight spill; it is loaded where it
// is used instead (WUSE; WREQ is empty) -- the product behind it is four times longer, the exposed round trip matters less.
#define WREQ(reg, ptr) do { if (D == 64) wfrag_t<D>(reg, ptr, strip, lane); } while (0)
#define WUSE(reg, ptr) do { if (D != 64) wfrag_t<D>(reg, ptr, strip, lane); } while (0)

struct SeEmbed {
    const float *E, *P;   // item table [R, D] (row 0 = padding), position table [S, D]; E == nullptr: x0 is given
    int64_t R;
    float scale;
};

template <int D, bool TRAIN>
__global__ __launch_bounds__(512) void enc_fwd_k(const float* __restrict__ x0, SeEmbed em, const int64_t* __restrict__ seq, int B, int S, int L,
                                                 SasrecParams P, float drop_scale, uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                 float* __restrict__ tape, EncTape T, const void* __restrict__ planp, int fill_pads,
                                                 const uint32_t* __restrict__ seed_dev) {
    using C = EC<D>;
    constexpr int KPT = C::KPT;             // keys per thread in the softmax phase
    if (seed_dev) seed ^= seed_dev[0];      // per-step seed kept in device memory (hipGraph replays)
    extern __shared__ __align__(16) float lds[];
    float* bX = lds;
    float* bA = bX + C::BUF;
    float* bQ = bA + C::BUF;
    float* bK = bQ + C::BUF;
    float* bV = bK + C::BUF;
    float* sP = bV + C::BUF;
    float* bK0 = sP + C::PBUF;                 // prefix k / v tiles of a chained part (allocated only where parts can chain: MAXT < 4)
    float* bV0 = bK0 + C::BUF;
    __shared__ int s_gid[C::ROWS], s_first[C::ROWS], s_pad[C::ROWS], s_sid[C::ROWS];
    __shared__ float s_w[C::ROWS];
    __shared__ float s_par[2 * EP_NPAR * D], s_last[2 * D];

    const int tid0 = threadIdx.x;
    const float inv_sqrt_d = 1.0f / sqrtf((float)D);
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_items = PL.hdr[0];

    for (int k = 0; k * (int)gridDim.x < n_items; ++k) {
        const int wi = enc_item_of(k, blockIdx.x, gridDim.x);
        if (wi >= n_items) continue;
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
        const int nsub = C::MAXT < 4 ? (whole.nt + C::MAXT - 1) / C::MAXT : 1;   // (D = 64 holds every sequence: no chaining code at all)
        for (int hs = 0; hs < nsub; ++hs) {
        const EncItem it = EncItem{whole.tile0 + hs * C::MAXT, whole.nt - hs * C::MAXT < C::MAXT ? whole.nt - hs * C::MAXT : C::MAXT, whole.kind};
        const int npre = C::MAXT < 4 ? hs * C::MAXT : 0;     // prefix key tiles
        const int64_t prow0 = (int64_t)whole.tile0 * 16;     // compact row of the sequence's first row
        if (hs > 0) __syncthreads();                         // (a full barrier: the previous part's tape stores have completed)
        const int nrows = 16 * it.nt;
        const int64_t row0 = (int64_t)it.tile0 * 16;
        int mk = 0; (void)mk;
        // block 0's small parameters and first three weight fragments are requested before anything else of the item
        ParRegs<D> PR;
        float wa[D / 4], wb[D / 4], wc[D / 4];
        par_fetch<D>(PR, P.blk[0], tid);
        const float lastv = tid < 2 * D ? (tid < D ? P.last_w[tid] : P.last_b[tid - D]) : 0.f;
        WREQ(wa, P.blk[0].in_w);
        WREQ(wb, P.blk[0].in_w + D * D);
        WREQ(wc, P.blk[0].in_w + 2 * D * D);
        enc_sync();
        ENC_MARK(g_fwd_marks, mk); ++mk;
        enc_decode<D>(PL, it, seq, tid, s_gid, s_first, s_pad);
        enc_sync();
        ENC_MARK(g_fwd_marks, mk); ++mk;
        if (tid < C::ROWS) s_sid[tid] = s_gid[tid] >= 0 ? s_gid[tid] / S : -1;
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

        for (int l = 0; l < L; ++l) {
            float* tp = TRAIN ? tape + (int64_t)l * T.per_block : nullptr;
            const float* par = s_par + (l & 1) * EP_NPAR * D;
            const bool more = l + 1 < L;
            const SasrecBlockParams Wn = P.blk[more ? l + 1 : l];   // the NEXT block's weights: requested two or more phases before use
            if (more) par_fetch<D>(PR, Wn, tid);
            TileRegs<D> TK0, TV0;
            if (C::MAXT < 4 && npre) {   // this block's k, v of the prefix rows (tape)
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
                WUSE(wa, W.in_w);
                gemm_rows<D>(bA, wa, lane, wr, it.nt, [&](int row, float v) { bQ[row * C::LS + col] = v + bq; });
                WREQ(wa, W.out_w);                 // Wo
                WUSE(wb, W.in_w + D * D);
                gemm_rows<D>(bX, wb, lane, wr, it.nt, [&](int row, float v) { bK[row * C::LS + col] = v + bk; });
                WREQ(wb, W.w1);                    // W1
                WUSE(wc, W.in_w + 2 * D * D);
                gemm_rows<D>(bX, wc, lane, wr, it.nt, [&](int row, float v) { bV[row * C::LS + col] = v + bv; });
                WREQ(wc, W.w2);                    // W2
                if (TRAIN) tile_store<D>(bA, tp + T.off_A + row0 * D, nrows, tid);
                if (C::MAXT < 4 && npre) {
                    tile_commit<D>(bK0, TK0, 16 * npre, tid);
                    tile_commit<D>(bV0, TV0, 16 * npre, tid);
                }
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 3. scores = q k^T / sqrt(D) over the item's (row tile, key tile) pairs
            gemm_pairs<D>(bQ, bK, lane, wave, it, [&](int row, int key, float v) { sP[row * C::PLS + key] = v * inv_sqrt_d; }, bK0, npre);
            if (TRAIN) {
                tile_store<D>(bQ, tp + T.off_Q + row0 * D, nrows, tid);
                tile_store<D>(bK, tp + T.off_K + row0 * D, nrows, tid);
                tile_store<D>(bV, tp + T.off_V + row0 * D, nrows, tid);
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- softmax over the keys of the same sequence with j <= i (causal; explicit pad rows ARE keys), plus the virtual
            //      pad key in front of the sequence (multiplicity first, score q.b_k/sqrt(D), value b_v); dropout on the probabilities
            if (r_e < nrows) {
                const int i = r_e;
                const int gi = s_gid[i], sid = s_sid[i], n_out = s_first[i];
                const int klo = 16 * enc_kt_lo(it, i >> 4), kpre = 16 * npre;   // key columns [0, kpre): prefix rows; own keys follow
                float p[KPT];
                float mx = -INFINITY;
                unsigned okm = 0;
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) {
                    const int j = j0_e + jj, jo = j - kpre;
                    const bool ok = gi >= 0 && (jo < 0 || (jo <= i && jo >= klo && s_sid[jo & (C::ROWS - 1)] == sid));
                    okm |= (ok ? 1u : 0u) << jj;
                    const float sv = sP[i * C::PLS + j];
                    p[jj] = ok ? sv : -INFINITY;
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
                    p[jj] = (p[jj] == -INFINITY) ? 0.f : expf(p[jj] - mx);
                    sum += p[jj];
                }
                sum = row_sum<C::TPR>(sum);
                const float epad = (spad == -INFINITY) ? 0.f : expf(spad - mx);
                sum += (float)n_out * epad;
                const float inv = (gi >= 0) ? 1.0f / sum : 0.f;
                const float ppad = epad * inv;
                float kept = (float)n_out;
                {   // each of the n_out pad keys has its own dropout bit (element (b, s_i, jj))
                    int cnt = 0;
                    if (thresh && gi >= 0)
                        for (int jj = (tid % C::TPR); jj < n_out; jj += C::TPR)
                            cnt += re_keep(seed, RE_STREAM_ATTN(l), (uint32_t)((int64_t)gi * S + jj), thresh) ? 1 : 0;
                    cnt = row_sum_i<C::TPR>(cnt);
                    if (thresh) kept = (float)cnt * drop_scale;
                }
                const float wv = (gi >= 0) ? ppad * kept : 0.f;
                if (row_lead) {
                    s_w[i] = wv;
                    if (TRAIN) {
                        float* pp = tp + T.off_PP + (row0 + i) * 2;
                        pp[0] = ppad; pp[1] = wv;
                    }
                }
                if (TRAIN) {   // pre-dropout probabilities (0 outside the row's window)
                    float* dst = tp + T.off_P + (row0 + i) * EP_PW + j0_e;
                    if (KPT % 4 == 0) {
#pragma unroll
                        for (int q = 0; q < KPT / 4; ++q)
                            reinterpret_cast<float4*>(dst)[q] = make_float4(p[4 * q] * inv, p[4 * q + 1] * inv, p[4 * q + 2] * inv, p[4 * q + 3] * inv);
                    } else {
#pragma unroll
                        for (int q = 0; q < KPT; ++q) dst[q] = p[q] * inv;
                    }
                }
                const int sbase = sid * S;
#pragma unroll
                for (int jj = 0; jj < KPT; ++jj) {
                    const int j = j0_e + jj;
                    float pr = p[jj] * inv;
                    if (((okm >> jj) & 1u) && thresh && pr != 0.f) {
                        // position of key column j inside the sequence (one long sequence: its rows are consecutive positions)
                        const int sj = it.kind ? n_out + j : s_gid[j & (C::ROWS - 1)] - sbase;
                        pr = re_keep(seed, RE_STREAM_ATTN(l), (uint32_t)((int64_t)gi * S + sj), thresh) ? pr * drop_scale : 0.f;
                    }
                    sP[i * C::PLS + j] = pr;
                }
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
                WUSE(wa, P.blk[l].out_w);
                gemm_rows<D>(bA, wa, lane, wr, it.nt, [&](int row, float v) { bQ[row * C::LS + col] = v + bo + bX[row * C::LS + col]; });
                if (more) WREQ(wa, Wn.in_w);                   // next block's Wq
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
                WUSE(wb, P.blk[l].w1);
                gemm_rows<D>(bK, wb, lane, wr, it.nt, [&](int row, float v) {
                    v += b1;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[row] * D + col);
                        v = re_keep(seed, RE_STREAM_FFN1(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    bV[row * C::LS + col] = fmaxf(v, 0.f);
                });
                if (more) WREQ(wb, Wn.in_w + D * D);           // next block's Wk
                if (TRAIN) tile_store<D>(bK, tp + T.off_Y + row0 * D, nrows, tid);
            }
            enc_sync();
            ENC_MARK(g_fwd_marks, mk); ++mk;
            // ---- 8. x' = dropout2(hr W2^T + b2) + y, pad rows zeroed
            {
                const float b2 = par[9 * D + col];
                WUSE(wc, P.blk[l].w2);
                gemm_rows<D>(bV, wc, lane, wr, it.nt, [&](int row, float v) {
                    v += b2;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[row] * D + col);
                        v = re_keep(seed, RE_STREAM_FFN2(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    v += bK[row * C::LS + col];
                    bX[row * C::LS + col] = s_pad[row] ? 0.f : v;
                });
                if (more) WREQ(wc, Wn.in_w + 2 * D * D);       // next block's Wv
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
            }
        }
        if (TRAIN) tile_store<D>(bX, tape + T.off_XL + row0 * D, nrows, tid);
        enc_sync();
        ENC_MARK(g_fwd_marks, mk); ++mk;
        tile_store_gid<D>(bA, u, s_gid, nrows, tid);
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
    }
}

extern "C" size_t re_sasrec_tape_bytes(int64_t B, int64_t S, int64_t D, int64_t L) {
    if (B <= 0 || S <= 0 || D <= 0 || L <= 0) return 256;
    return (size_t)enc_tape_layout(B, S, D, L).total * sizeof(float);
}

template <int D>
static int enc_fwd_launch_d(const float* x0, const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P,
                            float ds, uint32_t thresh, uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan,
                            int grid, int fill_pads, hipStream_t s) {
    using C = EC<D>;
    const EncTape T = enc_tape_layout(B, S, D, L);
    const size_t ldsb = (size_t)(5 * C::BUF + C::PBUF + (C::MAXT < 4 ? 2 * C::BUF : 0)) * sizeof(float);
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
