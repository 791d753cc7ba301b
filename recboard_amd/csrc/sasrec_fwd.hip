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

template <bool TRAIN>
__global__ __launch_bounds__(256) void sasrec_encoder_fwd_k(const float* __restrict__ x0, const int64_t* __restrict__ seq,
                                                            int B, int S, int L, SasrecParams P, float drop_scale,
                                                            uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                            float* __restrict__ tape, SasrecTape T) {
    extern __shared__ __align__(16) float lds[];
    float* bX = lds;
    float* bA = bX + SE_BUF;
    float* bQ = bA + SE_BUF;
    float* bK = bQ + SE_BUF;
    float* bV = bK + SE_BUF;
    float* bP = bV + SE_BUF;
    __shared__ int s_pad[SE_ROWS];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15, col = 16 * wave + c;
    const int r_e = tid >> 2, c0_e = (tid & 3) * 16;  // element-wise mapping
    const float inv_sqrt_d = 0.125f;                  // 1/sqrt(64)
    const int64_t SD = (int64_t)S * SE_D;

    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        __syncthreads();
        tile_load(bX, x0 + (int64_t)b * SD, S, tid);
        if (tid < SE_ROWS) s_pad[tid] = (tid < S) ? (seq[(int64_t)b * S + tid] == 0) : 1;
        __syncthreads();

        for (int l = 0; l < L; ++l) {
            const SasrecBlockParams W = se_launder(P.blk[l]);
            float* tp = TRAIN ? tape + (int64_t)l * T.per_block : nullptr;
            // ---- 1. Q-input = LN_a(x)
            {
                float mean, rstd;
                ln_row(bX, bA, W.ln_a_w, W.ln_a_b, tid, mean, rstd);
                if (TRAIN) {
                    tile_store(bX, tp + T.off_X + (int64_t)b * SD, S, tid);
                    if ((tid & 3) == 0 && r_e < S) {
                        float* st = tp + T.off_SA + ((int64_t)b * S + r_e) * 2;
                        st[0] = mean; st[1] = rstd;
                    }
                }
            }
            __syncthreads();
            // ---- 2. q, k, v projections
            {
                float bf[16];
                wfrag_kc(bf, W.in_w, wave, lane);
                const float bq = W.in_b[col];
                gemm64<true>(bA, bf, lane, [&](int row, float v) { bQ[row * SE_LS + col] = v + bq; });
                wfrag_kc(bf, W.in_w + SE_D * SE_D, wave, lane);
                const float bk = W.in_b[SE_D + col];
                gemm64<true>(bX, bf, lane, [&](int row, float v) { bK[row * SE_LS + col] = v + bk; });
                wfrag_kc(bf, W.in_w + 2 * SE_D * SE_D, wave, lane);
                const float bv = W.in_b[2 * SE_D + col];
                gemm64<true>(bX, bf, lane, [&](int row, float v) { bV[row * SE_LS + col] = v + bv; });
            }
            __syncthreads();
            if (TRAIN) {
                tile_store(bQ, tp + T.off_Q + (int64_t)b * SD, S, tid);
                tile_store(bK, tp + T.off_K + (int64_t)b * SD, S, tid);
                tile_store(bV, tp + T.off_V + (int64_t)b * SD, S, tid);
            }
            // ---- 3. scores = q k^T / sqrt(D)   (B^T = K, k-contiguous in LDS)
            {
                float bf[16];
                frag_kc(bf, bK + (16 * wave + c) * SE_LS + 16 * g);
                gemm64<true>(bQ, bf, lane, [&](int row, float v) { bP[row * SE_LS + col] = v * inv_sqrt_d; });
            }
            __syncthreads();
            // ---- softmax over keys j <= i (causal), rows >= S and columns > i are zero; dropout on the probabilities
            {
                const int i = r_e;
                float p[16];
                float mx = -INFINITY;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int j = c0_e + jj;
                    p[jj] = (j <= i && i < S) ? bP[i * SE_LS + j] : -INFINITY;
                    mx = fmaxf(mx, p[jj]);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
                mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
                float sum = 0.f;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    p[jj] = (p[jj] == -INFINITY) ? 0.f : expf(p[jj] - mx);
                    sum += p[jj];
                }
                sum = quad_sum(sum);
                const float inv = (i < S) ? 1.0f / sum : 0.f;
#pragma unroll
                for (int jj = 0; jj < 16; ++jj) {
                    const int j = c0_e + jj;
                    float pr = p[jj] * inv;
                    if (TRAIN && i < S && j < S) tp[T.off_P + ((int64_t)b * S + i) * S + j] = pr;
                    if (thresh && pr != 0.f) {
                        const uint32_t e = (uint32_t)(((int64_t)b * S + i) * S + j);
                        pr = re_keep(seed, RE_STREAM_ATTN(l), e, thresh) ? pr * drop_scale : 0.f;
                    }
                    bP[i * SE_LS + j] = pr;
                }
            }
            __syncthreads();
            // ---- 4. o = A v   (B[k=j][n=d] = V[j][d]: k strided)
            {
                float bf[16];
                frag_ks(bf, bV + (16 * g) * SE_LS + col, SE_LS);
                gemm64<true>(bP, bf, lane, [&](int row, float v) { bA[row * SE_LS + col] = v; });
            }
            __syncthreads();
            if (TRAIN) tile_store(bA, tp + T.off_O + (int64_t)b * SD, S, tid);
            // ---- 5. x1 = o Wo^T + bo + x
            {
                float bf[16];
                wfrag_kc(bf, W.out_w, wave, lane);
                const float bo = W.out_b[col];
                gemm64<true>(bA, bf, lane, [&](int row, float v) { bQ[row * SE_LS + col] = v + bo + bX[row * SE_LS + col]; });
            }
            __syncthreads();
            // ---- 6. y = LN_f(x1)
            {
                float mean, rstd;
                ln_row(bQ, bK, W.ln_f_w, W.ln_f_b, tid, mean, rstd);
                if (TRAIN) {
                    tile_store(bQ, tp + T.off_X1 + (int64_t)b * SD, S, tid);
                    if ((tid & 3) == 0 && r_e < S) {
                        float* st = tp + T.off_SF + ((int64_t)b * S + r_e) * 2;
                        st[0] = mean; st[1] = rstd;
                    }
                }
            }
            __syncthreads();
            // ---- 7. hr = relu(dropout1(y W1^T + b1))
            {
                float bf[16];
                wfrag_kc(bf, W.w1, wave, lane);
                const float b1 = W.b1[col];
                gemm64<true>(bK, bf, lane, [&](int row, float v) {
                    v += b1;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((((int64_t)b * S + row) * SE_D) + col);
                        v = re_keep(seed, RE_STREAM_FFN1(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    bV[row * SE_LS + col] = fmaxf(v, 0.f);
                });
            }
            __syncthreads();
            if (TRAIN) tile_store(bV, tp + T.off_HR + (int64_t)b * SD, S, tid);
            // ---- 8. x' = dropout2(hr W2^T + b2) + y, pad rows zeroed
            {
                float bf[16];
                wfrag_kc(bf, W.w2, wave, lane);
                const float b2 = W.b2[col];
                gemm64<true>(bV, bf, lane, [&](int row, float v) {
                    v += b2;
                    if (thresh) {
                        const uint32_t e = (uint32_t)((((int64_t)b * S + row) * SE_D) + col);
                        v = re_keep(seed, RE_STREAM_FFN2(l), e, thresh) ? v * drop_scale : 0.f;
                    }
                    v += bK[row * SE_LS + col];
                    bX[row * SE_LS + col] = s_pad[row] ? 0.f : v;
                });
            }
            __syncthreads();
        }
        // ---- u = LN_last(x_L)
        {
            float mean, rstd;
            ln_row(bX, bA, se_launder(P.last_w), se_launder(P.last_b), tid, mean, rstd);
            if (TRAIN) {
                tile_store(bX, tape + T.off_XL + (int64_t)b * SD, S, tid);
                if ((tid & 3) == 0 && r_e < S) {
                    float* st = tape + T.off_SL + ((int64_t)b * S + r_e) * 2;
                    st[0] = mean; st[1] = rstd;
                }
            }
        }
        __syncthreads();
        tile_store(bA, u + (int64_t)b * SD, S, tid);
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

extern "C" int re_sasrec_encoder_fwd(const float* x0, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                                     const float* const* block_params, const float* last_w, const float* last_b, float drop_p,
                                     uint32_t seed, float* u, void* tape, size_t tape_bytes, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!x0 || !seq || !u || B < 0) return RE_EINVAL;
    if (D != SE_D || S < 1 || S > SE_ROWS || L > SE_MAX_BLOCKS) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    SasrecParams P;
    if (!se_fill_params(P, block_params, L, last_w, last_b)) return RE_EINVAL;
    const SasrecTape T = sasrec_tape_layout(B, S, D, L);
    if (tape && tape_bytes < (size_t)T.total * sizeof(float)) return RE_EWORKSPACE;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const size_t ldsb = (size_t)6 * SE_BUF * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
    const int grid = (int)(B < 2048 ? B : 2048);
    if (tape) {
        auto k = sasrec_encoder_fwd_k<true>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, s, x0, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)tape, T);
    } else {
        auto k = sasrec_encoder_fwd_k<false>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), ldsb, s, x0, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)nullptr, T);
    }
    return re_launch_status();
}
