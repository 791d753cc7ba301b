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
