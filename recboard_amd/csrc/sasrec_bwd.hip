// K6/K7 backward: one launch per SASRec block (l = L-1 .. 0), one sequence per workgroup iteration, everything in LDS.
//
// Gradient of the block of sasrec_fwd.hip (SASRec/main.py:163-176 + :31-50) w.r.t. its input and its 12 parameters,
// reading the forward's tape (x, q, k, v, P, o, x1, relu(h), LN statistics) instead of recomputing the forward:
// 16 GEMMs of 64^3 per sequence per block (2 per linear map: dX = dY W and dW += dY^T X).  Layer-normed operands
// (LN_a(x), LN_f(x1)) are rebuilt from the saved statistics; dropout masks are regenerated from (seed, stream, index).
//
// Parameter gradients: each wave keeps its SE_RT 16x16 tiles of the six 64x64 weight gradients in registers (24 VGPRs
// at 16 waves) across all the work items its persistent workgroup processes; bias / LayerNorm gradients are per-thread
// column partials.  At the end every workgroup writes ONE slab (SB_SLAB floats); sasrec_grad_reduce sums the slabs in a
// fixed order straight into the gradient tensors: deterministic, no float atomics (cdna_hip_programming.md G12).
//
// MFMA-bound: 16 GEMMs x 2*64^3 = 8.39 MFLOP per sequence per block.
#include <math.h>

#ifndef SE_NW_BWD
#define SE_NW_BWD 8
#endif
#define SE_NW SE_NW_BWD
#include "sasrec_common.h"

#define SB_NMAT 6
#define SB_NVEC 12
#define SB_OFFP (SB_NMAT * 4096 + SB_NVEC * 64)   // position-table gradient slots (block 0 with the embedding backward fused in)
#define SB_SLAB (SB_OFFP + 4096)
// matrix slots: 0 Wq 1 Wk 2 Wv 3 Wo 4 W1 5 W2;  vector slots: 0 bq 1 bk 2 bv 3 bo 4 b1 5 b2 6 ga 7 ba 8 gf 9 bf 10 glast 11 blast

template <bool A_KC>
__device__ __forceinline__ void gemm64_acc(const float* A, const float (&bf)[16], int lane, int wr, f32x4 (&acc)[SE_RT]) {
    const int g = lane >> 4, c = lane & 15;
    float af[SE_RT][16];
#pragma unroll
    for (int t = 0; t < SE_RT; ++t) {
        const int tt = wr * SE_RT + t;
        if (A_KC) frag_kc(af[t], A + SE_RO(16 * tt + c) + 16 * g);
        else frag_ks(af[t], A + SE_RO(16 * g) + 16 * tt + c, SE_LS);
    }
    __builtin_amdgcn_sched_barrier(0);   // (see gemm64: fragment loads first, then the MFMAs back to back)
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
        for (int t = 0; t < SE_RT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][s], bf[s], acc[t], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
}

// column sums: thread (column tid & 63, row group tid >> 6) covers rows [SE_RPW * (tid >> 6), +SE_RPW)
__device__ __forceinline__ float colsum16(const float* tile, int tid) {
    const int c = tid & 63, r0 = (tid >> 6) * SE_RPW;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SE_RPW; ++i) s += tile[SE_RO(r0 + i) + c];
    return s;
}
// sum over the thread's rows of w[row] * tile[row][col]
__device__ __forceinline__ float colsum16_w(const float* tile, const float* w, int tid) {
    const int c = tid & 63, r0 = (tid >> 6) * SE_RPW;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SE_RPW; ++i) s = fmaf(w[r0 + i], tile[SE_RO(r0 + i) + c], s);
    return s;
}
// sum over the thread's rows of dy * xhat, xhat = (x - mean[row]) * rstd[row]
__device__ __forceinline__ float colsum16_xhat(const float* dy, const float* x, const float* mean, const float* rstd, int tid) {
    const int c = tid & 63, r0 = (tid >> 6) * SE_RPW;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SE_RPW; ++i) {
        const int r = r0 + i;
        s = fmaf(dy[SE_RO(r) + c], (x[SE_RO(r) + c] - mean[r]) * rstd[r], s);
    }
    return s;
}
// LayerNorm backward for this thread's row slice: dst (+)= rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * gamma
template <bool ACCUM>
__device__ __forceinline__ void ln_bwd_row(const float* dy, const float* x, float* dst, const float* __restrict__ gamma,
                                           const float* mean, const float* rstd, int tid) {
    const int r = tid / SE_TPR, c0 = (tid % SE_TPR) * SE_CPT;
    float d[SE_CPT], xv[SE_CPT];
    frag_row(d, dy + SE_RO(r) + c0);
    frag_row(xv, x + SE_RO(r) + c0);
    const float mu = mean[r], rs = rstd[r];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < SE_CPT; ++i) {
        xv[i] = (xv[i] - mu) * rs;
        d[i] *= gamma[c0 + i];
        s1 += d[i];
        s2 = fmaf(d[i], xv[i], s2);
    }
    s1 = row_sum(s1) * (1.0f / SE_D);
    s2 = row_sum(s2) * (1.0f / SE_D);
#pragma unroll
    for (int i = 0; i < SE_CPT; ++i) {
        const float v = rs * (d[i] - s1 - xv[i] * s2);
        if (ACCUM) dst[SE_RO(r) + c0 + i] += v; else dst[SE_RO(r) + c0 + i] = v;
    }
}
// y = (x - mean) * rstd * gamma + beta from saved statistics
__device__ __forceinline__ void ln_apply_row(const float* x, float* dst, const float* __restrict__ gw, const float* __restrict__ gb,
                                             const float* mean, const float* rstd, int tid) {
    const int r = tid / SE_TPR, c0 = (tid % SE_TPR) * SE_CPT;
    const float mu = mean[r], rs = rstd[r];
#pragma unroll
    for (int i = 0; i < SE_CPT; ++i) dst[SE_RO(r) + c0 + i] = (x[SE_RO(r) + c0 + i] - mu) * rs * gw[c0 + i] + gb[c0 + i];
}

// (two-step form of load_stats: request early, commit when the arrays are free)
__device__ __forceinline__ void stats_fetch(float2& r, const float* __restrict__ st, const int* s_gid, int tid) {
    r = make_float2(0.f, 0.f);
    if (tid < SE_ROWS) {
        const int gid = s_gid[tid];
        if (gid >= 0) r = *reinterpret_cast<const float2*>(st + 2 * (int64_t)gid);
    }
}
__device__ __forceinline__ void stats_commit(float* s_a, float* s_b, const float2& r, int tid) {
    if (tid < SE_ROWS) { s_a[tid] = r.x; s_b[tid] = r.y; }
}
__device__ __forceinline__ void load_stats(float* s_a, float* s_b, const float* __restrict__ st, const int* s_gid, int tid) {
    if (tid < SE_ROWS) {
        const int gid = s_gid[tid];
        s_a[tid] = gid >= 0 ? st[2 * (int64_t)gid] : 0.f;
        s_b[tid] = gid >= 0 ? st[2 * (int64_t)gid + 1] : 0.f;
    }
}

#ifdef SE_PROFILE
extern "C" int re_dbg_encoder_marks_bwd(unsigned long long* out64) {
    return hipMemcpyFromSymbol(out64, HIP_SYMBOL(g_se_marks), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
#endif

template <bool FIRST>
__global__ __launch_bounds__(SE_NT) void sasrec_block_bwd_k(const float* __restrict__ dIn, const int64_t* __restrict__ seq, int B, int S,
                                                          int l, SasrecBlockParams W0, const float* __restrict__ last_w0,
                                                          float drop_scale, uint32_t thresh, uint32_t seed,
                                                          const float* __restrict__ tape, SasrecTape T,
                                                          float* __restrict__ dOut, float* __restrict__ slab,
                                                          const int* __restrict__ order, const int* __restrict__ nshort_ptr,
                                                          const uint32_t* __restrict__ seed_dev, int fuse_embed, float emb_scale) {
    if (seed_dev) seed ^= seed_dev[0];   // per-step seed kept in device memory (hipGraph replays)
    extern __shared__ __align__(16) float lds[];
    float* b0 = lds;
    float* b1 = b0 + SE_BUF;
    float* b2 = b1 + SE_BUF;
    float* b3 = b2 + SE_BUF;
    float* b4 = b3 + SE_BUF;
    float* b5 = b4 + SE_BUF;
    float* b6 = b5 + SE_BUF;
    float* bW0 = b6 + SE_BUF;   // two staged weight matrices (see wtile_fetch)
    float* bW1 = bW0 + SE_BUF;
    __shared__ float s_mean[SE_ROWS], s_rstd[SE_ROWS];
    __shared__ float s_ppad[SE_ROWS], s_w[SE_ROWS], s_cpad[SE_ROWS];   // virtual pad key: prob of one copy, total kept weight, dS
    __shared__ int s_gid[SE_ROWS], s_grp[SE_ROWS], s_pad[SE_ROWS];

    const int tid0 = threadIdx.x;
    const float* tp = tape + (int64_t)l * T.per_block;
    const SeWork WK = se_work(B, nshort_ptr);

    float accV[SB_NVEC];
#pragma unroll
    for (int v = 0; v < SB_NVEC; ++v) accV[v] = 0.f;
    float accP[SE_CPT];   // fuse_embed: this thread's slice of the position-table gradient, slot = position + (64 - S)
#pragma unroll
    for (int i = 0; i < SE_CPT; ++i) accP[i] = 0.f;

    for (int wi = blockIdx.x; wi < WK.total; wi += gridDim.x) {
        SE_THREAD_VARS(tid0);
        const SasrecBlockParams W = se_launder(W0);
        const float* last_w = se_launder(last_w0);
        __syncthreads();
        SE_MARK(1, 0);
        const int n_out = se_decode(wi, WK, B, S, order, seq, tid, s_gid, s_grp, s_pad);
        const bool packed = wi < WK.nsw;   // four short sequences: attention is block diagonal over the 16-row tiles
        // weight gradients go straight to this workgroup's slab (overwritten by its first work item, added to by later ones):
        // keeping the six 64x64 accumulators in registers across the item loop cost 48 VGPRs per wave for a loop that
        // almost always runs once (work items <= workgroups)
        float* slw = slab + ((int64_t)l * gridDim.x + blockIdx.x) * SB_SLAB;
        const bool first_item = wi == (int)blockIdx.x;
        auto slab_add = [&](int m, int row, float v) {
            float* p = slw + m * 4096 + row * SE_D + col;
            *p = first_item ? v : *p + v;
        };
        __syncthreads();
        SE_MARK(1, 1);
        float4 R[SE_WV];                 // the next weight matrix, in flight from global memory
        TileRegs T0, T1;                 // the next tape tiles, in flight from global memory (requested a phase ahead)
        float2 ST;                       // ... and the next LayerNorm statistics / pad-key pair
        wtile_fetch(R, W.w2, tid);
        tile_fetch(T0, dIn, s_gid, tid);
        tile_fetch(T1, FIRST ? tape + T.off_XL : tp + T.off_HR, s_gid, tid);
        if (FIRST) {
            stats_fetch(ST, tape + T.off_SL, s_gid, tid);
            stats_commit(s_mean, s_rstd, ST, tid);
        }
        tile_commit(b0, T0, tid);
        tile_commit(b1, T1, tid);
        if (FIRST) tile_fetch(T1, tp + T.off_HR, s_gid, tid);
        tile_fetch(T0, tp + T.off_X1, s_gid, tid);
        stats_fetch(ST, tp + T.off_SF, s_gid, tid);
        __syncthreads();
        if (FIRST) {  // u = LN_last(x_L): dgamma/dbeta, then dx_L in place
            accV[10] += colsum16_xhat(b0, b1, s_mean, s_rstd, tid);
            accV[11] += colsum16(b0, tid);
            __syncthreads();
            ln_bwd_row<false>(b0, b1, b0, last_w, s_mean, s_rstd, tid);
            __syncthreads();
            tile_commit(b1, T1, tid);    // HR
        }
        SE_MARK(1, 2);
        // ---- pad mask of the block output (x'[pad] = 0) and dO2 = dX' * dropout2 mask
        {
            const int r = r_e, c0 = c0_e;
            const bool dead = s_pad[r] != 0;
#pragma unroll
            for (int i = 0; i < SE_CPT; ++i) {
                float v = dead ? 0.f : b0[SE_RO(r) + c0 + i];
                b0[SE_RO(r) + c0 + i] = v;
                if (thresh) {
                    const uint32_t e = (uint32_t)((int64_t)s_gid[r] * SE_D + c0 + i);
                    v = re_keep(seed, RE_STREAM_FFN2(l), e, thresh) ? v * drop_scale : 0.f;
                }
                b2[SE_RO(r) + c0 + i] = v;
            }
        }
        wtile_commit(bW0, R, tid);       // W2
        wtile_fetch(R, W.w1, tid);
        __syncthreads();
        SE_MARK(1, 3);
        // ---- A. FFN second map: dW2 += dO2^T hr; db2; dH = (dO2 W2) * (hr > 0) * scale
        {
            float bf[16];
            wtile_commit(bW1, R, tid);   // W1
            wtile_fetch(R, W.out_w, tid);
            frag_ks(bf, b1 + SE_RO(16 * g) + col, SE_LS);
            gemm64<false>(b2, bf, lane, wr, [&](int row, float v) { slab_add(5, row, v); });
            accV[5] += colsum16(b2, tid);
            wtile_frag_n(bf, bW0, wc, lane);
            gemm64<true>(b2, bf, lane, wr, [&](int row, float v) {
                b3[SE_RO(row) + col] = (b1[SE_RO(row) + col] > 0.f) ? v * drop_scale : 0.f;
            });
        }
        __syncthreads();
        SE_MARK(1, 4);
        // ---- B. FFN first map: y = LN_f(x1) rebuilt; dW1 += dH^T y; db1; dY = dH W1 + dX'
        tile_commit(b1, T0, tid);        // X1
        stats_commit(s_mean, s_rstd, ST, tid);
        tile_fetch(T0, tp + T.off_O, s_gid, tid);
        __syncthreads();
        ln_apply_row(b1, b2, W.ln_f_w, W.ln_f_b, s_mean, s_rstd, tid);
        __syncthreads();
        {
            float bf[16];
            wtile_commit(bW0, R, tid);   // Wo  (bW0's last reader, GEMM A, finished before phase A's closing barrier)
            wtile_fetch(R, W.in_w, tid);
            frag_ks(bf, b2 + SE_RO(16 * g) + col, SE_LS);
            gemm64<false>(b3, bf, lane, wr, [&](int row, float v) { slab_add(4, row, v); });
            accV[4] += colsum16(b3, tid);
            wtile_frag_n(bf, bW1, wc, lane);
            gemm64<true>(b3, bf, lane, wr, [&](int row, float v) { b0[SE_RO(row) + col] += v; });
        }
        __syncthreads();
        SE_MARK(1, 5);
        // ---- C. LN_f backward: dgamma_f, dbeta_f, dX1 (in place in b0)
        accV[8] += colsum16_xhat(b0, b1, s_mean, s_rstd, tid);
        accV[9] += colsum16(b0, tid);
        __syncthreads();
        ln_bwd_row<false>(b0, b1, b0, W.ln_f_w, s_mean, s_rstd, tid);
        __syncthreads();
        SE_MARK(1, 6);
        // ---- D. out_proj: dWo += dX1^T o; dbo; dO = dX1 Wo
        tile_commit(b2, T0, tid);        // O
        tile_fetch(T0, tp + T.off_V, s_gid, tid);
        stats_fetch(ST, tp + T.off_PP, s_gid, tid);
        float pq[SE_CPT];                // this thread's slice of the saved probabilities
        {
            const int gi = s_gid[r_e];
#pragma unroll
            for (int q = 0; q < SE_CPT / 4; ++q) {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gi >= 0) v = reinterpret_cast<const float4*>(tp + T.off_P + (int64_t)gi * SE_ROWS + c0_e)[q];
                pq[4 * q] = v.x; pq[4 * q + 1] = v.y; pq[4 * q + 2] = v.z; pq[4 * q + 3] = v.w;
            }
        }
        __syncthreads();
        {
            float bf[16];
            wtile_commit(bW1, R, tid);   // Wq  (bW1's last reader, GEMM B, finished before phase B's closing barrier)
            wtile_fetch(R, W.in_w + SE_D * SE_D, tid);
            frag_ks(bf, b2 + SE_RO(16 * g) + col, SE_LS);
            gemm64<false>(b0, bf, lane, wr, [&](int row, float v) { slab_add(3, row, v); });
            accV[3] += colsum16(b0, tid);
            wtile_frag_n(bf, bW0, wc, lane);
            gemm64<true>(b0, bf, lane, wr, [&](int row, float v) { b3[SE_RO(row) + col] = v; });
        }
        __syncthreads();
        SE_MARK(1, 7);
        // ---- E. attention: load V, P; Pd = P*mask; dP = (dO V^T)*mask; dV = Pd^T dO; dS = P (dP - rowsum(dP P)) / sqrt(D)
        tile_commit(b1, T0, tid);        // V
        stats_commit(s_ppad, s_w, ST, tid);   // (p_pad, w) of the virtual out-of-window pad key
        tile_fetch(T0, tp + T.off_Q, s_gid, tid);
        tile_fetch(T1, tp + T.off_K, s_gid, tid);
        {
            const int i = r_e, j0 = c0_e;
            const int gi = s_gid[i];
            const int sbase = gi >= 0 ? (gi / S) * S : 0;
#pragma unroll
            for (int jj = 0; jj < SE_CPT; ++jj) {
                const int j = j0 + jj;
                const float p = pq[jj];          // 0 outside the causal / same-sequence window (the forward stored zeros there)
                float m = 1.0f;
                if (thresh && p != 0.f) {
                    const int sj = s_gid[j] - sbase;
                    m = re_keep(seed, RE_STREAM_ATTN(l), (uint32_t)((int64_t)gi * S + sj), thresh) ? drop_scale : 0.f;
                }
                b4[SE_RO(i) + j] = p;
                b6[SE_RO(i) + j] = p * m;
            }
        }
        __syncthreads();
        SE_MARK(1, 8);
        {
            float bf[16];
            frag_kc(bf, b1 + SE_RO(16 * wc + c) + 16 * g);  // B^T = V (k = d contiguous)
            gemm64<true>(b3, bf, lane, wr, [&](int row, float v) {
                // grad w.r.t. the pre-dropout probability: scale by the same mask factor Pd/P (0, or 1/(1-p))
                const float p = b4[SE_RO(row) + col];
                const float pd = b6[SE_RO(row) + col];
                b5[SE_RO(row) + col] = (p != 0.f) ? v * (pd / p) : 0.f;
            }, packed ? wc : -1);   // (packed: the off-diagonal tiles of dP are neither computed nor read -- P is 0 there)
            if (packed) {
                gemm64_diag<false>(b6, b3, lane, wr, col, [&](int row, float v) { b2[SE_RO(row) + col] = v; });  // dV
            } else {
                frag_ks(bf, b3 + SE_RO(16 * g) + col, SE_LS);     // B[k=i][n=d] = dO
                gemm64<false>(b6, bf, lane, wr, [&](int row, float v) { b2[SE_RO(row) + col] = v; });  // dV
            }
        }
        if (n_out > 0) accV[2] += colsum16_w(b3, s_w, tid);     // d b_v through the virtual pad key: sum_i w_i dO_i
        __syncthreads();
        SE_MARK(1, 9);
        {
            const int i = r_e, j0 = c0_e;
            float dp[SE_CPT], pp[SE_CPT];
            frag_row(dp, b5 + SE_RO(i) + j0);
            frag_row(pp, b4 + SE_RO(i) + j0);
            float s = 0.f;
#pragma unroll
            for (int jj = 0; jj < SE_CPT; ++jj) {
                if (pp[jj] == 0.f) dp[jj] = 0.f;   // (never-written dP entries of a packed item's off-diagonal tiles)
                s = fmaf(dp[jj], pp[jj], s);
            }
            s = row_sum(s);
            if (n_out > 0) {
                // virtual pad key: upstream grad of each copy = (dO_i . b_v) * mask; t = dO_i . b_v
                float t = 0.f;
#pragma unroll
                for (int jj = 0; jj < SE_CPT; ++jj) t = fmaf(b3[SE_RO(i) + j0 + jj], W.in_b[2 * SE_D + j0 + jj], t);
                t = row_sum(t);
                const float wv = s_w[i], ppad = s_ppad[i];
                s = fmaf(t, wv, s);                                            // rowdot includes the pad copies
                if (row_lead) s_cpad[i] = (wv * t - (float)n_out * ppad * s) * 0.125f;   // sum of dS over the copies
            }
#pragma unroll
            for (int jj = 0; jj < SE_CPT; ++jj) b5[SE_RO(i) + j0 + jj] = pp[jj] * (dp[jj] - s) * 0.125f;
        }
        __syncthreads();   // dS complete; dO (b3), V (b1) and P (b4) no longer needed
        SE_MARK(1, 10);
        // ---- F. dQ = dS K -> b4 ; dK = dS^T Q -> b6
        tile_commit(b3, T0, tid);        // Q
        tile_commit(b1, T1, tid);        // K
        tile_fetch(T0, tp + T.off_X, s_gid, tid);
        stats_fetch(ST, tp + T.off_SA, s_gid, tid);
        __syncthreads();
        {
            float bf[16];
            wtile_commit(bW0, R, tid);   // Wk  (bW0's last reader, GEMM D, is several barriers back)
            wtile_fetch(R, W.in_w + 2 * SE_D * SE_D, tid);
            const float bkc = (n_out > 0) ? W.in_b[SE_D + col] : 0.f;
            if (packed) {
                gemm64_diag<true>(b5, b1, lane, wr, col, [&](int row, float v) {
                    b4[SE_RO(row) + col] = (n_out > 0) ? fmaf(s_cpad[row], bkc, v) : v;   // + dS_pad * b_k
                });
                gemm64_diag<false>(b5, b3, lane, wr, col, [&](int row, float v) { b6[SE_RO(row) + col] = v; });
            } else {
                frag_ks(bf, b1 + SE_RO(16 * g) + col, SE_LS);
                gemm64<true>(b5, bf, lane, wr, [&](int row, float v) {
                    b4[SE_RO(row) + col] = (n_out > 0) ? fmaf(s_cpad[row], bkc, v) : v;   // + dS_pad * b_k
                });
                frag_ks(bf, b3 + SE_RO(16 * g) + col, SE_LS);
                gemm64<false>(b5, bf, lane, wr, [&](int row, float v) { b6[SE_RO(row) + col] = v; });
            }
            if (n_out > 0) accV[1] += colsum16_w(b3, s_cpad, tid);   // d b_k through the virtual pad key: sum_i dS_pad_i q_i
        }
        __syncthreads();
        SE_MARK(1, 11);
        // ---- G. projections: x, LN_a(x) rebuilt; dWq/dWk/dWv, biases; dA1 = dQ Wq -> b5; dX (b0) += dK Wk + dV Wv
        tile_commit(b1, T0, tid);        // X
        stats_commit(s_mean, s_rstd, ST, tid);
        __syncthreads();
        ln_apply_row(b1, b3, W.ln_a_w, W.ln_a_b, s_mean, s_rstd, tid);
        __syncthreads();
        {
            float bf[16];
            frag_ks(bf, b3 + SE_RO(16 * g) + col, SE_LS);     // LN_a(x)
            gemm64<false>(b4, bf, lane, wr, [&](int row, float v) { slab_add(0, row, v); });
            frag_ks(bf, b1 + SE_RO(16 * g) + col, SE_LS);     // x
            gemm64<false>(b6, bf, lane, wr, [&](int row, float v) { slab_add(1, row, v); });
            gemm64<false>(b2, bf, lane, wr, [&](int row, float v) { slab_add(2, row, v); });
            accV[0] += colsum16(b4, tid);
            accV[1] += colsum16(b6, tid);
            accV[2] += colsum16(b2, tid);
            wtile_frag_n(bf, bW1, wc, lane);                   // Wq
            gemm64<true>(b4, bf, lane, wr, [&](int row, float v) { b5[SE_RO(row) + col] = v; });
            wtile_frag_n(bf, bW0, wc, lane);                   // Wk
            gemm64<true>(b6, bf, lane, wr, [&](int row, float v) { b0[SE_RO(row) + col] += v; });
            __syncthreads();                                   // every wave is done with Wq
            wtile_commit(bW1, R, tid);                         // Wv
            __syncthreads();
            wtile_frag_n(bf, bW1, wc, lane);
            gemm64<true>(b2, bf, lane, wr, [&](int row, float v) { b0[SE_RO(row) + col] += v; });
        }
        __syncthreads();
        SE_MARK(1, 12);
        // ---- H. LN_a backward: dgamma_a, dbeta_a; dX += LN_a'(dA1)
        accV[6] += colsum16_xhat(b5, b1, s_mean, s_rstd, tid);
        accV[7] += colsum16(b5, tid);
        ln_bwd_row<true>(b5, b1, b0, W.ln_a_w, s_mean, s_rstd, tid);
        __syncthreads();
        SE_MARK(1, 13);
        if (fuse_embed) {
            // re_sasrec_embed_bwd fused in (block 0): pad rows -> 0, the embedding's dropout mask, position-table gradient
            // (unscaled sum over the batch), and the rows go out scaled by sqrt(D) as item-gradient contributions
            {
                const bool dead = s_pad[r_e] != 0;
#pragma unroll
                for (int i = 0; i < SE_CPT; ++i) {
                    float v = dead ? 0.f : b0[SE_RO(r_e) + c0_e + i];
                    if (thresh && !dead) {
                        const uint32_t e = (uint32_t)((int64_t)s_gid[r_e] * SE_D + c0_e + i);
                        v = re_keep(seed, RE_STREAM_EMBED, e, thresh) ? v * drop_scale : 0.f;
                    }
                    b0[SE_RO(r_e) + c0_e + i] = v;
                }
            }
            __syncthreads();
            if (wi < WK.nsw) {   // packed item: rows 16q + i of the four sequences all sit at position S - 16 + i  (slot 48 + i)
                if (r_e >= SE_ROWS - SE_WIN) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int i = 0; i < SE_CPT; ++i) accP[i] += b0[SE_RO(16 * q + r_e - (SE_ROWS - SE_WIN)) + c0_e + i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < SE_CPT; ++i) accP[i] += b0[SE_RO(r_e) + c0_e + i];
            }
            for (int f = tid; f < SE_ROWS * (SE_D / 4); f += SE_NT) {
                const int r = f >> 4, c4 = f & 15;
                const int gid = s_gid[r];
                if (gid >= 0) {
                    const float4 v = *reinterpret_cast<const float4*>(b0 + SE_RO(r) + 4 * c4);
                    reinterpret_cast<float4*>(dOut + (int64_t)gid * SE_D)[c4] = make_float4(v.x * emb_scale, v.y * emb_scale, v.z * emb_scale, v.w * emb_scale);
                }
            }
        } else {
            tile_store(b0, dOut, s_gid, tid);
        }
        SE_MARK(1, 14);
    }

    // ---- this workgroup's slab (workgroups without a work item write nothing: the reduction only reads the first
    //      min(work items, gridDim.x) slabs of the block's region)
    if ((int)blockIdx.x >= WK.total) return;
    SE_THREAD_VARS(tid0);
    float* sl = slab + ((int64_t)l * gridDim.x + blockIdx.x) * SB_SLAB;
    if (fuse_embed) {
#pragma unroll
        for (int i = 0; i < SE_CPT; ++i) sl[SB_OFFP + r_e * SE_D + c0_e + i] = accP[i];
    }
    // column partials of the SE_NW row groups -> one value per column, added in row-group order
    __syncthreads();
    float* red = lds;   // [SB_NVEC][SE_NW][64]
#pragma unroll
    for (int v = 0; v < SB_NVEC; ++v) red[(v * SE_NW + wave) * 64 + lane] = accV[v];
    __syncthreads();
    for (int e = tid; e < SB_NVEC * 64; e += SE_NT) {
        const int v = e >> 6, cc = e & 63;
        float s = red[(v * SE_NW) * 64 + cc];
#pragma unroll
        for (int i = 1; i < SE_NW; ++i) s += red[(v * SE_NW + i) * 64 + cc];
        sl[SB_NMAT * 4096 + v * 64 + cc] = s;
    }
}

struct SasrecGradDst {
    float* p[SE_MAX_BLOCKS][14];  // per block: ABI order of the 12 block gradients, then g_last_w, g_last_b (last block only)
};

// Slab reduction in two fixed-order levels (deterministic), ONE launch each for all blocks (blockIdx.z / .y = block):
// level 1 sums groups of SB_RGROUP slabs with one thread per (group, element) -- SB_RGROUP independent loads in flight per
// thread -- level 2 adds the group partials in order and writes the gradient tensors.  Only the slabs of workgroups that
// had a work item exist: nact = min(work items, nwg), derived on the device from the packing metadata.
#define SB_RGROUP 16
__global__ __launch_bounds__(256) void sasrec_slab_partial(const float* __restrict__ slab, int nwg, int ngroups, float* __restrict__ part,
                                                           int B, const int* __restrict__ nshort_ptr) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= SB_SLAB) return;
    const int total = se_work(B, nshort_ptr).total;
    const int nact = total < nwg ? total : nwg;
    const int w0 = blockIdx.y * SB_RGROUP;
    if (w0 >= nact) return;
    const float* sl = slab + (int64_t)blockIdx.z * nwg * SB_SLAB;
    float v[SB_RGROUP];
#pragma unroll
    for (int i = 0; i < SB_RGROUP; ++i) v[i] = (w0 + i < nact) ? sl[(int64_t)(w0 + i) * SB_SLAB + e] : 0.f;
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SB_RGROUP; ++i) s += v[i];
    part[((int64_t)blockIdx.z * ngroups + blockIdx.y) * SB_SLAB + e] = s;
}

__global__ __launch_bounds__(256) void sasrec_grad_reduce(const float* __restrict__ part, int nwg, int ngroups, SasrecGradDst dst, int L,
                                                          int B, const int* __restrict__ nshort_ptr, float* __restrict__ dP, int S) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const int l = blockIdx.y;
    const int NM = SB_NMAT * 4096;
    const int nvec = (l == L - 1) ? SB_NVEC : SB_NVEC - 2;
    if (e >= SB_SLAB) return;
    const bool is_p = e >= SB_OFFP;
    if (is_p) {
        if (l != 0 || !dP || ((e - SB_OFFP) >> 6) < SE_ROWS - S) return;   // slots in front of position 0 are unused
    } else if (e >= NM + nvec * 64) {
        return;
    }
    const int total = se_work(B, nshort_ptr).total;
    const int nact = total < nwg ? total : nwg;
    const int ng = (nact + SB_RGROUP - 1) / SB_RGROUP;
    const float* pl = part + (int64_t)l * ngroups * SB_SLAB;
    float s = 0.f;
    for (int w = 0; w < ng; ++w) s += pl[(int64_t)w * SB_SLAB + e];
    float* const* P = dst.p[l];
    float* d;
    if (is_p) {
        d = dP + (e - SB_OFFP) - (SE_ROWS - S) * SE_D;
    } else if (e < NM) {
        const int m = e >> 12, off = e & 4095;
        d = ((m < 3) ? P[2] + m * 4096 : (m == 3 ? P[4] : (m == 4 ? P[8] : P[10]))) + off;
    } else {
        const int v = (e - NM) >> 6, cc = (e - NM) & 63;
        switch (v) {
            case 0: case 1: case 2: d = P[3] + v * 64; break;
            case 3: d = P[5]; break;
            case 4: d = P[9]; break;
            case 5: d = P[11]; break;
            case 6: d = P[0]; break;
            case 7: d = P[1]; break;
            case 8: d = P[6]; break;
            case 9: d = P[7]; break;
            case 10: d = P[12]; break;
            default: d = P[13]; break;
        }
        d += cc;
    }
    *d = s;
}

#define SB_MAX_WGS 256

extern "C" size_t re_sasrec_encoder_bwd_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t L) {
    if (B <= 0) return 256;
    const int64_t nwg = B < SB_MAX_WGS ? B : SB_MAX_WGS;
    const int64_t ngroups = (nwg + SB_RGROUP - 1) / SB_RGROUP;
    return (size_t)(L * (nwg + ngroups) * SB_SLAB + 2 * B * S * D) * sizeof(float) + 512;
}

static int se_bwd_launch(const float* dU, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                         const float* const* block_params, const float* last_w, const float* last_b, float drop_p, uint32_t seed,
                         const uint32_t* seed_dev, const void* tape, float* dx0, float* const* block_grads, float* g_last_w,
                         float* g_last_b, void* ws, size_t ws_bytes, const int32_t* order, const int32_t* nshort, float emb_scale,
                         float* dP, re_stream_t stream) {
    if (B == 0) return RE_OK;
    if (!dU || !seq || !tape || !dx0 || !block_params || !block_grads || !g_last_w || !g_last_b || !last_w || !last_b || !ws || B < 0)
        return RE_EINVAL;
    if (D != SE_D || S < 1 || S > SE_ROWS || L < 1 || L > SE_MAX_BLOCKS) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    if ((order == nullptr) != (nshort == nullptr)) return RE_EINVAL;
    if (ws_bytes < re_sasrec_encoder_bwd_workspace_bytes(B, S, D, L)) return RE_EWORKSPACE;
    for (int64_t i = 0; i < 12 * L; ++i)
        if (!block_params[i] || !block_grads[i]) return RE_EINVAL;
    const SasrecTape T = sasrec_tape_layout(B, S, D, L);
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int nwg = (int)(B < SB_MAX_WGS ? B : SB_MAX_WGS);
    const int ngroups = (nwg + SB_RGROUP - 1) / SB_RGROUP;
    float* slab = (float*)ws;
    float* part = slab + (size_t)L * nwg * SB_SLAB;
    float* dxa = part + (size_t)L * ngroups * SB_SLAB;
    float* dxb = dxa + (size_t)B * S * D;
    const size_t ldsb = (size_t)9 * SE_BUF * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    auto kf = sasrec_block_bwd_k<true>;
    auto kn = sasrec_block_bwd_k<false>;
    static bool attr_done = false;   // benign race: the attribute is idempotent; kept out of captured regions after the first call
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        if (hipFuncSetAttribute((const void*)kn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        attr_done = true;
    }
    const float* din = dU;
    for (int64_t l = L - 1; l >= 0; --l) {
        const float* const* q = block_params + 12 * l;
        SasrecBlockParams W{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]};
        float* dout = (l == 0) ? dx0 : (((L - 1 - l) & 1) ? dxb : dxa);
        const bool first = (l == L - 1);
        const int fuse = (l == 0 && dP) ? 1 : 0;
        if (first)
            hipLaunchKernelGGL(kf, dim3(nwg), dim3(SE_NT), ldsb, s, din, seq, (int)B, (int)S, (int)l, W, last_w, ds, thresh, seed,
                               (const float*)tape, T, dout, slab, order, nshort, seed_dev, fuse, emb_scale);
        else
            hipLaunchKernelGGL(kn, dim3(nwg), dim3(SE_NT), ldsb, s, din, seq, (int)B, (int)S, (int)l, W, last_w, ds, thresh, seed,
                               (const float*)tape, T, dout, slab, order, nshort, seed_dev, fuse, emb_scale);
        din = dout;
    }
    SasrecGradDst dst;
    for (int64_t l = 0; l < SE_MAX_BLOCKS; ++l)
        for (int i = 0; i < 14; ++i) dst.p[l][i] = (l < L && i < 12) ? block_grads[12 * l + i] : (i == 12 ? g_last_w : g_last_b);
    hipLaunchKernelGGL(sasrec_slab_partial, dim3((SB_SLAB + 255) / 256, ngroups, (unsigned)L), dim3(256), 0, s, slab, nwg, ngroups, part, (int)B,
                       nshort);
    hipLaunchKernelGGL(sasrec_grad_reduce, dim3((SB_SLAB + 255) / 256, (unsigned)L), dim3(256), 0, s, part, nwg, ngroups, dst, (int)L, (int)B,
                       nshort, dP, (int)S);
    return re_launch_status();
}

extern "C" int re_sasrec_encoder_bwd(const float* dU, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                                     const float* const* block_params, const float* last_w, const float* last_b, float drop_p,
                                     uint32_t seed, const uint32_t* seed_dev, const void* tape, float* dx0, float* const* block_grads,
                                     float* g_last_w,
                                     float* g_last_b, void* ws, size_t ws_bytes, const int32_t* order, const int32_t* nshort,
                                     re_stream_t stream) {
    re_clear_error();
    return se_bwd_launch(dU, seq, B, S, D, L, block_params, last_w, last_b, drop_p, seed, seed_dev, tape, dx0, block_grads, g_last_w, g_last_b,
                         ws, ws_bytes, order, nshort, 0.f, nullptr, stream);
}

extern "C" int re_sasrec_encoder_embed_bwd(const float* dU, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                                           const float* const* block_params, const float* last_w, const float* last_b, float drop_p,
                                           uint32_t seed, const uint32_t* seed_dev, const void* tape, float scale, float* contrib,
                                           float* dP, float* const* block_grads, float* g_last_w, float* g_last_b, void* ws,
                                           size_t ws_bytes, const int32_t* order, const int32_t* nshort, re_stream_t stream) {
    re_clear_error();
    if (B != 0 && !dP) return RE_EINVAL;
    return se_bwd_launch(dU, seq, B, S, D, L, block_params, last_w, last_b, drop_p, seed, seed_dev, tape, contrib, block_grads, g_last_w,
                         g_last_b, ws, ws_bytes, order, nshort, scale, dP, stream);
}
