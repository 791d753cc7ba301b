// K6/K7 forward: the whole SASRec encoder (all blocks + lastLN) for one sequence per workgroup, activations in LDS.
//
// Reference restated: SASRec/main.py:163-176 (after_one_block), :31-50 (PointWiseFeedForward), :188-191.
//   q = LN_a(x) Wq^T + bq;  k = x Wk^T + bk;  v = x Wv^T + bv          (K,V are NOT layer-normed)
//   A = dropout(softmax(q k^T / sqrt(D) + causal));  x1 = (A v) Wo^T + bo + x   (pad positions ARE attended as keys)
//   y = LN_f(x1);  x' = dropout2(relu(dropout1(y W1^T + b1)) W2^T + b2) + y;  x'[pad] = 0
//   u = LN_last(x_L)
// The reference runs ~30 aten kernels per block through HBM; here a sequence's 12.8 KB of activations never leave
// the CU between the embedding and lastLN, weights (100 KB/block, shared by every workgroup) stream from L2 as MFMA
// B-fragments, and -- in training -- each intermediate the backward needs is written once to the tape.
//
// MFMA-bound: 8 GEMMs of 64^3 per block per sequence = 4.19 MFLOP (x L blocks).
#include <math.h>

#include "sasrec_common.h"

#ifdef SE_PROFILE
extern "C" int re_dbg_encoder_marks_fwd(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_se_marks), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#endif

// The encoder's input tile straight from the tables (re_sasrec_embed fused in):  x0 = seq == 0 ? 0 : dropout(E[seq] * scale + P[s])
struct SeEmbed {
    const float *E, *P;   // item table [R, 64] (row 0 = padding), position table [S, 64]; E == nullptr: x0 is given
    int64_t R;
    float scale;
};
__device__ __forceinline__ void tile_embed(float* tile, const SeEmbed& em, const int64_t* __restrict__ seq, int S, float drop_scale,
                                           uint32_t thresh, uint32_t seed, const int* s_gid, int tid) {
    for (int f = tid; f < SE_ROWS * (SE_D / 4); f += SE_NT) {
        const int r = f >> 4, c4 = f & 15;
        const int gid = s_gid[r];
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gid >= 0) {
            const int64_t item = seq[gid];
            if (item > 0 && item < em.R) {
                const float4 v = reinterpret_cast<const float4*>(em.E + item * SE_D)[c4];
                const float4 p = reinterpret_cast<const float4*>(em.P + (int64_t)(gid % S) * SE_D)[c4];
                o.x = v.x * em.scale + p.x;
                o.y = v.y * em.scale + p.y;
                o.z = v.z * em.scale + p.z;
                o.w = v.w * em.scale + p.w;
                if (thresh) {
                    const uint32_t e = (uint32_t)((int64_t)gid * SE_D + c4 * 4);
                    o.x = re_keep(seed, RE_STREAM_EMBED, e + 0, thresh) ? o.x * drop_scale : 0.f;
                    o.y = re_keep(seed, RE_STREAM_EMBED, e + 1, thresh) ? o.y * drop_scale : 0.f;
                    o.z = re_keep(seed, RE_STREAM_EMBED, e + 2, thresh) ? o.z * drop_scale : 0.f;
                    o.w = re_keep(seed, RE_STREAM_EMBED, e + 3, thresh) ? o.w * drop_scale : 0.f;
                }
            }
        }
        *reinterpret_cast<float4*>(tile + SE_RO(r) + 4 * c4) = o;
    }
}

template <bool TRAIN>
__global__ __launch_bounds__(SE_NT) void sasrec_encoder_fwd_k(const float* __restrict__ x0, SeEmbed em, const int64_t* __restrict__ seq,
                                                            int B, int S, int L, SasrecParams P, float drop_scale,
                                                            uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                            float* __restrict__ tape, SasrecTape T,
                                                            const int* __restrict__ order, const int* __restrict__ nshort_ptr,
                                                            int fill_pads, const uint32_t* __restrict__ seed_dev) {
    if (seed_dev) seed ^= seed_dev[0];   // per-step seed kept in device memory (hipGraph replays)
    extern __shared__ __align__(16) float lds[];
    float* bX = lds;
    float* bA = bX + SE_BUF;
    float* bQ = bA + SE_BUF;
    float* bK = bQ + SE_BUF;
    float* bV = bK + SE_BUF;
    float* bP = bV + SE_BUF;
    float* bW0 = bP + SE_BUF;   // two staged weight matrices (see wtile_fetch)
    float* bW1 = bW0 + SE_BUF;
    __shared__ int s_gid[SE_ROWS], s_grp[SE_ROWS], s_pad[SE_ROWS];
    __shared__ float s_w[SE_ROWS];

    const int tid0 = threadIdx.x;
    const float inv_sqrt_d = 0.125f;                  // 1/sqrt(64)
    const SeWork WK = se_work(B, nshort_ptr);

    for (int wi = blockIdx.x; wi < WK.total; wi += gridDim.x) {
        SE_THREAD_VARS(tid0);
        __syncthreads();
        SE_MARK(0, 0);
        const int n_out = se_decode(wi, WK, B, S, order, seq, tid, s_gid, s_grp, s_pad);
        const bool packed = wi < WK.nsw;   // four short sequences: attention is block diagonal over the 16-row tiles
        __syncthreads();
        if (em.E) tile_embed(bX, em, seq, S, drop_scale, thresh, seed, s_gid, tid);
        else tile_load(bX, x0, s_gid, tid);
        __syncthreads();
        SE_MARK(0, 1);

        for (int l = 0; l < L; ++l) {
            SE_THREAD_VARS(tid0);
            const SasrecBlockParams W = se_launder(P.blk[l]);
            float* tp = TRAIN ? tape + (int64_t)l * T.per_block : nullptr;
            float4 R[SE_WV];                  // the next weight matrix, in flight from global memory
            wtile_fetch(R, W.in_w, tid);      // Wq
            // ---- 1. Q-input = LN_a(x)
            {
                float mean, rstd;
                ln_row(bX, bA, W.ln_a_w, W.ln_a_b, tid, mean, rstd);
                if (TRAIN) {
                    tile_store(bX, tp + T.off_X, s_gid, tid);
                    if (row_lead && s_gid[r_e] >= 0) {
                        float* st = tp + T.off_SA + (int64_t)s_gid[r_e] * 2;
                        st[0] = mean; st[1] = rstd;
                    }
                }
            }
            wtile_commit(bW0, R, tid);                              // Wq
            wtile_fetch(R, W.in_w + SE_D * SE_D, tid);              // Wk
            __syncthreads();
            if (l == 0) SE_MARK(0, 2);
            // ---- 2. q, k, v projections
            {
                const float bq = W.in_b[col], bk = W.in_b[SE_D + col], bv = W.in_b[2 * SE_D + col];
                float bf[16];
                wtile_commit(bW1, R, tid);                          // Wk
                wtile_fetch(R, W.in_w + 2 * SE_D * SE_D, tid);      // Wv
                wtile_frag_t(bf, bW0, wc, lane);
                gemm64<true>(bA, bf, lane, wr, [&](int row, float v) { bQ[SE_RO(row) + col] = v + bq; });
                __syncthreads();
                wtile_commit(bW0, R, tid);                          // Wv
                wtile_fetch(R, W.out_w, tid);                       // Wo
                wtile_frag_t(bf, bW1, wc, lane);
                gemm64<true>(bX, bf, lane, wr, [&](int row, float v) { bK[SE_RO(row) + col] = v + bk; });
                __syncthreads();
                wtile_frag_t(bf, bW0, wc, lane);
                gemm64<true>(bX, bf, lane, wr, [&](int row, float v) { bV[SE_RO(row) + col] = v + bv; });
            }
            __syncthreads();
            if (TRAIN) {
                tile_store(bQ, tp + T.off_Q, s_gid, tid);
                tile_store(bK, tp + T.off_K, s_gid, tid);
                tile_store(bV, tp + T.off_V, s_gid, tid);
            }
            if (l == 0) SE_MARK(0, 3);
            // ---- 3. scores = q k^T / sqrt(D)   (B^T = K, k-contiguous in LDS)
            {
                float bf[16];
                frag_kc(bf, bK + SE_RO(16 * wc + c) + 16 * g);
                gemm64<true>(bQ, bf, lane, wr, [&](int row, float v) { bP[SE_RO(row) + col] = v * inv_sqrt_d; }, packed ? wc : -1);
            }
            __syncthreads();
            if (l == 0) SE_MARK(0, 4);
            // ---- softmax over the keys of the same sequence with j <= i (causal; pads ARE keys), plus the virtual
            //      out-of-window pad key (multiplicity n_out, score q.b_k/sqrt(D), value b_v); dropout on the probabilities
            {
                const int i = r_e;
                const int gi = s_gid[i], grp = s_grp[i];
                float p[SE_CPT];
                float mx = -INFINITY;
                unsigned okm = 0;
#pragma unroll
                for (int jj = 0; jj < SE_CPT; ++jj) {
                    const int j = c0_e + jj;
                    const bool ok = gi >= 0 && j <= i && s_gid[j] >= 0 && s_grp[j] == grp;
                    okm |= (ok ? 1u : 0u) << jj;
                    p[jj] = ok ? bP[SE_RO(i) + j] : -INFINITY;
                    mx = fmaxf(mx, p[jj]);
                }
                float spad = -INFINITY;
                if (n_out > 0) {  // workgroup-uniform
                    float d = 0.f;
#pragma unroll
                    for (int jj = 0; jj < SE_CPT; ++jj) d = fmaf(bQ[SE_RO(i) + c0_e + jj], W.in_b[SE_D + c0_e + jj], d);
                    spad = (gi >= 0) ? row_sum(d) * inv_sqrt_d : -INFINITY;
                    mx = fmaxf(mx, spad);
                }
                mx = row_max(mx);
                float sum = 0.f;
#pragma unroll
                for (int jj = 0; jj < SE_CPT; ++jj) {
                    p[jj] = (p[jj] == -INFINITY) ? 0.f : expf(p[jj] - mx);
                    sum += p[jj];
                }
                sum = row_sum(sum);
                const float epad = (spad == -INFINITY) ? 0.f : expf(spad - mx);
                sum += (float)n_out * epad;
                const float inv = (gi >= 0) ? 1.0f / sum : 0.f;
                if (n_out > 0) {
                    const float ppad = epad * inv;
                    float kept = (float)n_out;
                    if (thresh) {  // each of the n_out pad keys has its own dropout bit (element (b, s_i, jj))
                        int cnt = 0;
                        if (gi >= 0)
                            for (int jj = (tid % SE_TPR); jj < n_out; jj += SE_TPR)
                                cnt += re_keep(seed, RE_STREAM_ATTN(l), (uint32_t)((int64_t)gi * S + jj), thresh) ? 1 : 0;
                        cnt = row_sum_i(cnt);
                        kept = (float)cnt * drop_scale;
                    }
                    const float wv = (gi >= 0) ? ppad * kept : 0.f;
                    if (row_lead) {
                        s_w[i] = wv;
                        if (TRAIN && gi >= 0) {
                            float* pp = tp + T.off_PP + (int64_t)gi * 2;
                            pp[0] = ppad; pp[1] = wv;
                        }
                    }
                } else if (row_lead) {
                    s_w[i] = 0.f;
                }
                if (TRAIN && gi >= 0) {   // pre-dropout probabilities, tile-column layout: one aligned store per SE_CPT columns
#pragma unroll
                    for (int q = 0; q < SE_CPT / 4; ++q)
                        reinterpret_cast<float4*>(tp + T.off_P + (int64_t)gi * SE_ROWS + c0_e)[q] =
                            make_float4(p[4 * q] * inv, p[4 * q + 1] * inv, p[4 * q + 2] * inv, p[4 * q + 3] * inv);
                }
#pragma unroll
                for (int jj = 0; jj < SE_CPT; ++jj) {
                    const int j = c0_e + jj;
                    float pr = p[jj] * inv;
                    if ((okm >> jj) & 1u) {
                        const int sj = s_gid[j] - (gi / S) * S;   // position of key j inside the sequence
                        if (thresh && pr != 0.f)
                            pr = re_keep(seed, RE_STREAM_ATTN(l), (uint32_t)((int64_t)gi * S + sj), thresh) ? pr * drop_scale : 0.f;
                    }
                    bP[SE_RO(i) + j] = pr;
                }
            }
            __syncthreads();
            if (l == 0) SE_MARK(0, 5);
            // ---- 4. o = A v + w * b_v   (B[k=j][n=d] = V[j][d]: k strided)
            {
                float bf[16];
                wtile_commit(bW1, R, tid);                          // Wo
                wtile_fetch(R, W.w1, tid);                          // W1
                const float bv = W.in_b[2 * SE_D + col];
                if (packed) {
                    gemm64_diag<true>(bP, bV, lane, wr, col, [&](int row, float v) { bA[SE_RO(row) + col] = fmaf(s_w[row], bv, v); });
                } else {
                    frag_ks(bf, bV + SE_RO(16 * g) + col, SE_LS);
                    gemm64<true>(bP, bf, lane, wr, [&](int row, float v) { bA[SE_RO(row) + col] = fmaf(s_w[row], bv, v); });
                }
            }
            __syncthreads();
            if (TRAIN) tile_store(bA, tp + T.off_O, s_gid, tid);
            if (l == 0) SE_MARK(0, 6);
            // ---- 5. x1 = o Wo^T + bo + x
            {
                float bf[16];
                wtile_commit(bW0, R, tid);                          // W1
                wtile_fetch(R, W.w2, tid);                          // W2
                wtile_frag_t(bf, bW1, wc, lane);
                const float bo = W.out_b[col];
                gemm64<true>(bA, bf, lane, wr, [&](int row, float v) { bQ[SE_RO(row) + col] = v + bo + bX[SE_RO(row) + col]; });
            }
            __syncthreads();
            if (l == 0) SE_MARK(0, 7);
            // ---- 6. y = LN_f(x1)
            {
                float mean, rstd;
                ln_row(bQ, bK, W.ln_f_w, W.ln_f_b, tid, mean, rstd);
                if (TRAIN) {
                    tile_store(bQ, tp + T.off_X1, s_gid, tid);
                    if (row_lead && s_gid[r_e] >= 0) {
                        float* st = tp + T.off_SF + (int64_t)s_gid[r_e] * 2;
                        st[0] = mean; st[1] = rstd;
                    }
                }
            }
            __syncthreads();
            if (l == 0) SE_MARK(0, 8);
            // ---- 7. hr = relu(dropout1(y W1^T + b1))
            {
                float bf[16];
                wtile_commit(bW1, R, tid);                          // W2
                wtile_frag_t(bf, bW0, wc, lane);
                const float b1 = W.b1[col];
                gemm64<true>(bK, bf, lane, wr, [&](int row, float v) {
                    v += b1;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[row] * SE_D + col);
                        v = re_keep(seed, RE_STREAM_FFN1(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    bV[SE_RO(row) + col] = fmaxf(v, 0.f);
                });
            }
            __syncthreads();
            if (TRAIN) tile_store(bV, tp + T.off_HR, s_gid, tid);
            if (l == 0) SE_MARK(0, 9);
            // ---- 8. x' = dropout2(hr W2^T + b2) + y, pad rows zeroed
            {
                float bf[16];
                wtile_frag_t(bf, bW1, wc, lane);
                const float b2 = W.b2[col];
                gemm64<true>(bV, bf, lane, wr, [&](int row, float v) {
                    v += b2;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[row] * SE_D + col);
                        v = re_keep(seed, RE_STREAM_FFN2(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    v += bK[SE_RO(row) + col];
                    bX[SE_RO(row) + col] = s_pad[row] ? 0.f : v;
                });
            }
            __syncthreads();
        }
        SE_MARK(0, 10);
        // ---- u = LN_last(x_L)
        {
            float mean, rstd;
            ln_row(bX, bA, se_launder(P.last_w), se_launder(P.last_b), tid, mean, rstd);
            if (TRAIN) {
                tile_store(bX, tape + T.off_XL, s_gid, tid);
                if (row_lead && s_gid[r_e] >= 0) {
                    float* st = tape + T.off_SL + (int64_t)s_gid[r_e] * 2;
                    st[0] = mean; st[1] = rstd;
                }
            }
        }
        __syncthreads();
        tile_store(bA, u, s_gid, tid);
        SE_MARK(0, 11);
        if (fill_pads && n_out > 0) {
            // positions in front of the window are pads: u = LN_last(0) = beta_last (what the reference's encode returns there)
            for (int f = tid; f < 4 * n_out * (SE_D / 4); f += SE_NT) {
                const int c4 = f & 15, rr = f >> 4;
                const int t = rr / n_out, s = rr - t * n_out;
                const int q = 4 * wi + t;
                if (q < WK.nshort) {
                    const int b = order ? order[q] : q;
                    reinterpret_cast<float4*>(u + ((int64_t)b * S + s) * SE_D)[c4] = reinterpret_cast<const float4*>(P.last_b)[c4];
                }
            }
        }
    }
}

extern "C" size_t re_sasrec_tape_bytes(int64_t B, int64_t S, int64_t D, int64_t L) {
    if (B <= 0 || S <= 0 || D <= 0 || L <= 0) return 256;
    return (size_t)sasrec_tape_layout(B, S, D, L).total * sizeof(float);
}

static bool se_fill_params(SasrecParams& P, const float* const* bp, int64_t L, const float* last_w, const float* last_b) {
    if (!bp || !last_w || !last_b || L < 1 || L > SE_MAX_BLOCKS) return false;
    for (int64_t l = 0; l < L; ++l) {
        const float* const* q = bp + 12 * l;
        for (int i = 0; i < 12; ++i)
            if (!q[i]) return false;
        P.blk[l] = SasrecBlockParams{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]};
    }
    P.last_w = last_w;
    P.last_b = last_b;
    return true;
}

static int se_fwd_launch(const float* x0, const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                         const float* const* block_params, const float* last_w, const float* last_b, float drop_p, uint32_t seed,
                         const uint32_t* seed_dev, float* u, void* tape, size_t tape_bytes, const int32_t* order, const int32_t* nshort,
                         re_stream_t stream) {
    if (B == 0) return RE_OK;
    if ((!x0 && !em.E) || !seq || !u || B < 0) return RE_EINVAL;
    if (D != SE_D || S < 1 || S > SE_ROWS || L > SE_MAX_BLOCKS) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    SasrecParams P;
    if (!se_fill_params(P, block_params, L, last_w, last_b)) return RE_EINVAL;
    const SasrecTape T = sasrec_tape_layout(B, S, D, L);
    if (tape && tape_bytes < (size_t)T.total * sizeof(float)) return RE_EWORKSPACE;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const size_t ldsb = (size_t)8 * SE_BUF * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)(B < 2048 ? B : 2048);  // >= the number of work items (shorts packed 4 per item); idle blocks exit
    if ((order == nullptr) != (nshort == nullptr)) return RE_EINVAL;
    static bool attr_done[2] = {false, false};   // benign race: the attribute is idempotent
    if (tape) {
        auto k = sasrec_encoder_fwd_k<true>;
        if (!attr_done[1]) {
            if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
            attr_done[1] = true;
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(SE_NT), ldsb, s, x0, em, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)tape, T, order, nshort, 0, seed_dev);
    } else {
        auto k = sasrec_encoder_fwd_k<false>;
        if (!attr_done[0]) {
            if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
            attr_done[0] = true;
        }
        hipLaunchKernelGGL(k, dim3(grid), dim3(SE_NT), ldsb, s, x0, em, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)nullptr, T, order, nshort, 1, seed_dev);
    }
    return re_launch_status();
}

extern "C" int re_sasrec_encoder_fwd(const float* x0, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                                     const float* const* block_params, const float* last_w, const float* last_b, float drop_p,
                                     uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, size_t tape_bytes,
                                     const int32_t* order, const int32_t* nshort, re_stream_t stream) {
    re_clear_error();
    if (B != 0 && !x0) return RE_EINVAL;
    const SeEmbed em{nullptr, nullptr, 0, 0.f};
    return se_fwd_launch(x0, em, seq, B, S, D, L, block_params, last_w, last_b, drop_p, seed, seed_dev, u, tape, tape_bytes, order, nshort,
                         stream);
}

extern "C" int re_sasrec_embed_encoder_fwd(const float* E, int64_t R, const float* P, float scale, const int64_t* seq, int64_t B, int64_t S,
                                           int64_t D, int64_t L, const float* const* block_params, const float* last_w,
                                           const float* last_b, float drop_p, uint32_t seed, const uint32_t* seed_dev, float* u, void* tape,
                                           size_t tape_bytes, const int32_t* order, const int32_t* nshort, re_stream_t stream) {
    re_clear_error();
    if (B != 0 && (!E || !P || R <= 0)) return RE_EINVAL;
    if ((reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(P)) & 15u) return RE_EUNSUPPORTED;
    const SeEmbed em{E, P, R, scale};
    return se_fwd_launch(nullptr, em, seq, B, S, D, L, block_params, last_w, last_b, drop_p, seed, seed_dev, u, tape, tape_bytes, order, nshort,
                         stream);
}
