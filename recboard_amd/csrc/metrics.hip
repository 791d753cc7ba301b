// Ranking metrics from the fused top-K output (freerec.metrics through Coach.evaluate; contract mirrored at
// UniSRec/main.py:428-447: monitor(scores, targets, pool=[HITRATE, PRECISION, RECALL, NDCG, MRR]) for every NAME@K).
// The reference runs one torch.topk + gather over the dense [B, N] score matrix per "NAME@K" monitor (9 passes on the
// benchmark config); here every metric of every K comes from the ONE sorted top-Kmax list: one wave per user, lane =
// rank.  Definitions (textbook; parity unpinned, restated in oracle/ranking.py):
//   HITRATE@k = [any hit in top k], PRECISION@k = hits/k, RECALL@k = hits/#targets,
//   NDCG@k = DCG@k / IDCG@min(k, #targets) with gain 1/log2(rank+2), MRR@k = 1/(first hit rank + 1).
// out[b][ik][5]; a fixed-order two-stage sum over users gives the batch totals (Coach weights by n = batch size).
#include <math.h>

#include "re_common.h"

#define RM_MAXK 8

struct RankKs { int k[RM_MAXK]; int nk; };

__global__ __launch_bounds__(256) void rank_metrics_k(const int64_t* __restrict__ topk, int64_t B, int Kmax,
                                                      const int64_t* __restrict__ tgt_ptr, const int64_t* __restrict__ tgt_idx,
                                                      RankKs ks, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t t0 = tgt_ptr[b], t1 = tgt_ptr[b + 1];
    const int ntgt = (int)(t1 - t0);
    bool hit = false;
    if (lane < Kmax) {
        const int64_t id = topk[b * Kmax + lane];
        for (int64_t t = t0; t < t1; ++t) hit = hit || (id >= 0 && tgt_idx[t] == id);
    }
    const float gain = 1.0f / log2f((float)lane + 2.0f);
    const unsigned long long hm = __ballot(hit);
    for (int ik = 0; ik < ks.nk; ++ik) {
        const int k = ks.k[ik];
        const unsigned long long km = (k >= 64) ? ~0ull : ((1ull << k) - 1ull);
        const unsigned long long h = hm & km;
        const int nh = __popcll(h);
        float dcg = re_wave_sum((hit && lane < k) ? gain : 0.f);
        const int ni = ntgt < k ? (ntgt > 0 ? ntgt : 1) : k;
        float idcg = re_wave_sum(lane < ni ? gain : 0.f);
        if (lane == 0) {
            float* o = out + (b * ks.nk + ik) * 5;
            o[0] = nh > 0 ? 1.f : 0.f;
            o[1] = (float)nh / (float)k;
            o[2] = (float)nh / (float)(ntgt > 0 ? ntgt : 1);
            o[3] = dcg / idcg;
            o[4] = h ? 1.0f / (float)(__ffsll((long long)h)) : 0.f;
        }
    }
}

// sums[j] = sum_b out[b][j]  (j < nk*5): one block per j, fixed order
__global__ __launch_bounds__(256) void rank_metrics_sum(const float* __restrict__ per_user, int64_t B, int nj, float* __restrict__ sums) {
    __shared__ float s[4];
    const int j = blockIdx.x;
    float acc = 0.f;
    for (int64_t b = threadIdx.x; b < B; b += 256) acc += per_user[b * nj + j];
    acc = re_wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) sums[j] = ((s[0] + s[1]) + s[2]) + s[3];
}

extern "C" int re_rank_metrics(const int64_t* topk_idx, int64_t B, int64_t Kmax, const int64_t* tgt_ptr, const int64_t* tgt_idx,
                               const int32_t* h_ks, int32_t nk, float* per_user, float* sums, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!topk_idx || !tgt_ptr || !tgt_idx || !h_ks || !per_user || B < 0 || Kmax < 1 || Kmax > 64 || nk < 1 || nk > RM_MAXK) return RE_EINVAL;
    RankKs ks;
    ks.nk = nk;
    for (int i = 0; i < nk; ++i) {
        if (h_ks[i] < 1 || h_ks[i] > Kmax) return RE_EINVAL;
        ks.k[i] = h_ks[i];
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rank_metrics_k, dim3((unsigned)re_cdiv(B, 4)), dim3(256), 0, s, topk_idx, B, (int)Kmax, tgt_ptr, tgt_idx, ks, per_user);
    if (sums) hipLaunchKernelGGL(rank_metrics_sum, dim3(nk * 5), dim3(256), 0, s, (const float*)per_user, B, nk * 5, sums);
    return re_launch_status();
}

// ---- AUC of a prediction model's scores (DeepFM/configs/Frappe_x1_BARS.yaml:101-102 `monitors: [LOGLOSS, AUC]`, computed by
// freerec's Coach from `recommend_from_pool` outputs, DeepFM/main.py:217-219).  AUC is the Mann-Whitney statistic
//     AUC = ( #{(i, j): y_i = 1, y_j = 0, s_i > s_j} + 0.5 #{... s_i == s_j} ) / (P N)
// counted PAIRWISE: no sort, integer counts (exact, order-free -> deterministic), every (positive, negative) pair compared once.
// Thread = one sample i against a tile of samples j staged in LDS; grid = (i tiles) x (j tiles).  n = 30 000 evaluation rows are
// 9e8 comparisons; the count fits 64 bits for any n.  counts = { greater, equal, P, N } (uint64, zeroed by the launcher).
#define AUC_TILE 2048
__global__ __launch_bounds__(256) void auc_pairs_k(const float* __restrict__ s, const float* __restrict__ y, int64_t n,
                                                   unsigned long long* __restrict__ counts) {
    __shared__ float ls[AUC_TILE];
    __shared__ unsigned char lneg[AUC_TILE];
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * 256 + tid;
    const bool ipos = i < n && y[i] > 0.5f;
    const float si = i < n ? s[i] : 0.f;
    const int64_t j0 = (int64_t)blockIdx.y * AUC_TILE;
    for (int q = tid; q < AUC_TILE; q += 256) {
        const int64_t j = j0 + q;
        ls[q] = j < n ? s[j] : 0.f;
        lneg[q] = (j < n && !(y[j] > 0.5f)) ? 1 : 0;
    }
    __syncthreads();
    unsigned gt = 0, eq = 0;
    if (ipos) {
        for (int q = 0; q < AUC_TILE; ++q) {
            const float sj = ls[q];
            const unsigned m = lneg[q];
            gt += (si > sj) ? m : 0u;
            eq += (si == sj) ? m : 0u;
        }
    }
    // wave totals, one atomic pair per wave; the class counts are taken once (by the first column of j tiles)
    for (int o = 32; o > 0; o >>= 1) { gt += __shfl_xor(gt, o, 64); eq += __shfl_xor(eq, o, 64); }
    const unsigned long long mp = __ballot(ipos), mn = __ballot(i < n && !ipos);
    if ((tid & 63) == 0) {
        if (gt) atomicAdd(&counts[0], (unsigned long long)gt);
        if (eq) atomicAdd(&counts[1], (unsigned long long)eq);
        if (blockIdx.y == 0) {
            atomicAdd(&counts[2], (unsigned long long)__builtin_popcountll(mp));
            atomicAdd(&counts[3], (unsigned long long)__builtin_popcountll(mn));
        }
    }
}
__global__ void auc_final_k(const unsigned long long* __restrict__ counts, float* __restrict__ out) {
    const double pn = (double)counts[2] * (double)counts[3];
    out[0] = pn > 0.0 ? (float)(((double)counts[0] + 0.5 * (double)counts[1]) / pn) : 0.5f;
}

extern "C" size_t re_auc_workspace_bytes(void) { return 256; }
extern "C" int re_auc(const float* scores, const float* labels, int64_t n, float* auc, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (n < 0 || !auc || !ws || (n && (!scores || !labels))) return RE_EINVAL;
    if (ws_bytes < 32 || (reinterpret_cast<uintptr_t>(ws) & 7u)) return RE_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    if (re_zero_async(ws, 32, s) != hipSuccess) return RE_ELAUNCH;
    if (n > 0)
        hipLaunchKernelGGL(auc_pairs_k, dim3((unsigned)re_cdiv(n, 256), (unsigned)re_cdiv(n, AUC_TILE)), dim3(256), 0, s, scores, labels, n,
                           (unsigned long long*)ws);
    hipLaunchKernelGGL(auc_final_k, dim3(1), dim3(1), 0, s, (const unsigned long long*)ws, auc);
    return re_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// Pool ranking (`--ranking=pool`: recommend_from_pool, SASRec/main.py:230-236, MF-BPR/main.py:106-109, LightGCN/main.py:122-125; evaluate
// contract UniSRec/main.py:415-421: the target first, then the sampled unseen items; targets[:, 0] = 1).
//   re_score_pool: out[b][p] = <Q[b], E[pool[b][p]]> as the natural-k fmaf chain -- the value re_score_dense / re_score_topk give that
//                  (user, item) pair, bit for bit (einsum("BD,BKD->BK") of the reference).  A thread per pair; the pool's rows are a
//                  gather of B x P rows of 4 D bytes from a table that lives in L2 / Infinity Cache at these sizes.
//   re_pool_topk:  the exact top-K of every row of a [B, P] score matrix, ties to the lowest position (the full-ranking kernels' rule):
//                  one wave per row, a candidate's rank = the number of candidates that beat it (P^2 compares a row: P = 101).
__global__ __launch_bounds__(256) void score_pool_k(const float* __restrict__ Q, const float* __restrict__ E, const int64_t* __restrict__ pool,
                                                    int64_t B, int64_t P, int64_t N, int D, float* __restrict__ out) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= B * P) return;
    const int64_t b = e / P;
    const int64_t id = pool[e];
    if (id < 0 || id >= N) { out[e] = -INFINITY; return; }
    const float4* q = reinterpret_cast<const float4*>(Q + b * D);
    const float4* r = reinterpret_cast<const float4*>(E + id * D);
    float acc = 0.f;
    for (int k = 0; k < D / 4; ++k) {
        const float4 a = q[k], c = r[k];
        acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
    }
    out[e] = acc;
}

#define PT_MAXP 1024
__global__ __launch_bounds__(256) void pool_topk_k(const float* __restrict__ scores, int64_t B, int P, int K, float* __restrict__ vals,
                                                   int64_t* __restrict__ idx) {
    __shared__ float s[4][PT_MAXP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * 4 + w;
    if (b >= B) return;                                   // (whole waves; no workgroup barrier below)
    for (int i = lane; i < P; i += 64) s[w][i] = scores[b * P + i];
    for (int k = lane; k < K; k += 64)                    // slots beyond the pool: (-inf, -1), as re_score_topk pads K > N
        if (k >= P) { vals[b * K + k] = -INFINITY; idx[b * K + k] = -1; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int i = lane; i < P; i += 64) {
        const float si = s[w][i];
        int rank = 0;
        for (int j = 0; j < P; ++j) {
            const float sj = s[w][j];
            rank += (sj > si || (sj == si && j < i)) ? 1 : 0;
        }
        if (rank < K) { vals[b * K + rank] = si; idx[b * K + rank] = i; }
    }
}

extern "C" int re_score_pool(const float* Q, const float* E, const int64_t* pool, int64_t B, int64_t P, int64_t N, int64_t D, float* out,
                             re_stream_t stream) {
    re_clear_error();
    if (B == 0 || P == 0) return RE_OK;
    if (!Q || !E || !pool || !out || B < 0 || P < 0 || N <= 0 || D <= 0 || (D & 3)) return RE_EINVAL;
    hipLaunchKernelGGL(score_pool_k, dim3((unsigned)re_cdiv(B * P, 256)), dim3(256), 0, (hipStream_t)stream, Q, E, pool, B, P, N, (int)D, out);
    return re_launch_status();
}

extern "C" int re_pool_topk(const float* scores, int64_t B, int64_t P, int64_t K, float* vals, int64_t* idx, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!scores || !vals || !idx || B < 0 || P <= 0 || K <= 0) return RE_EINVAL;
    if (P > PT_MAXP) return RE_EUNSUPPORTED;
    hipLaunchKernelGGL(pool_topk_k, dim3((unsigned)re_cdiv(B, 4)), dim3(256), 0, (hipStream_t)stream, scores, B, (int)P, (int)K, vals, idx);
    return re_launch_status();
}
