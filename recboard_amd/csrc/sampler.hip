// Device-side batch assembly of the SASRec training chain:
//   shuffled_seqs_source(maxlen) -> seq_train_yielding_pos_(1, -1) -> seq_train_sampling_neg_(1) -> add_(1, (ISeq,)) -> lpad_(maxlen, ..., 0)
// (SASRec/main.py:143-157; row semantics HSTU/sampler.py:47-125): row b of a batch is user order[b0 + b] of the epoch's shuffled user
// list; with w = the last maxlen items of its training sequence (shuffled_seqs_source(maxlen) cuts the sequence BEFORE the target is split
// off: HSTU/sampler.py:28-31 `items[-maxlen:]`, so a row has at most maxlen - 1 inputs): ISeq = w[:-1] + 1, IPos = w[1:], both left-padded with 0 to
// maxlen; INeg = one uniform item per real position that is NOT in the user's training set (0 on pads).
// The training interactions live in HBM as two CSR arrays over users: `items` chronological, `sorted_items` ascending (the seen
// probe: a binary search per draw).  Draws come from the counter-based generator of re_rng.h keyed by (seed, step, position,
// attempt): the batch is a pure function of its arguments -- reproducible, no generator state, capturable.
// One thread per (row, position); HBM-latency-bound (three dependent loads per position), ~25 k positions per batch.
#include "re_common.h"
#include "re_rng.h"

#define RS_STREAM_NEG 0x5EEDu
#define RS_MAX_TRIES 32

__global__ __launch_bounds__(256) void seq_train_sample_k(const int64_t* __restrict__ ptr, const int64_t* __restrict__ items,
                                                          const int64_t* __restrict__ sorted_items, const int64_t* __restrict__ order,
                                                          int64_t n_order, int64_t b0, int B, int S, int64_t N, uint32_t seed, uint32_t step,
                                                          int64_t* __restrict__ users, int64_t* __restrict__ seq, int64_t* __restrict__ pos,
                                                          int64_t* __restrict__ neg) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S, s = i % S;
    int64_t vs = 0, vp = 0, vn = 0, u = -1;
    if (b0 + b < n_order) {
        u = order[b0 + b];
        const int64_t p0 = ptr[u], n = ptr[u + 1] - p0;
        const int64_t len = n - 1 < S - 1 ? n - 1 : S - 1;        // input positions of the row: window = the last min(n, S) items, one of them the last target
        const int64_t k = s - (S - len);
        if (len > 0 && k >= 0) {
            const int64_t base = p0 + n - 1 - len;
            vs = items[base + k] + 1;
            vp = items[base + k + 1];
            // a uniform item outside the user's training set: draw, probe the sorted list, redraw (the set is a sliver of the catalog:
            // the first draw passes almost always; after RS_MAX_TRIES the last draw stands, as a pathological user's would)
            const int64_t* sl = sorted_items + p0;
            const uint32_t ctr = (uint32_t)i * RS_MAX_TRIES;
            for (int t = 0; t < RS_MAX_TRIES; ++t) {
                const uint32_t r = re_rng_u32(seed ^ (step * 0x9E3779B1u), RS_STREAM_NEG, ctr + t);
                vn = (int64_t)(((uint64_t)r * (uint64_t)N) >> 32);
                int64_t lo = 0, hi = n;
                while (lo < hi) {
                    const int64_t mid = (lo + hi) >> 1;
                    if (sl[mid] < vn) lo = mid + 1; else hi = mid;
                }
                if (lo >= n || sl[lo] != vn) break;
            }
        }
    }
    seq[i] = vs; pos[i] = vp; neg[i] = vn;
    if (s == 0 && users) users[b] = u;
}

// MF-BPR / LightGCN training chain (MF-BPR/main.py:60-68: choiced_user_ids_source -> gen_train_sampling_pos_ -> gen_train_sampling_neg_(1)):
// row b = a uniformly drawn user among those with items (`order` = their ids), one of its training items, one unseen item.
__global__ __launch_bounds__(256) void gen_train_sample_k(const int64_t* __restrict__ ptr, const int64_t* __restrict__ items,
                                                          const int64_t* __restrict__ sorted_items, const int64_t* __restrict__ order,
                                                          int64_t n_order, int B, int64_t N, uint32_t seed, uint32_t step,
                                                          int64_t* __restrict__ users, int64_t* __restrict__ pos, int64_t* __restrict__ neg) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    const uint32_t key = seed ^ (step * 0x9E3779B1u);
    const uint32_t ctr = (uint32_t)b * (RS_MAX_TRIES + 2);
    const int64_t u = order[(int64_t)(((uint64_t)re_rng_u32(key, RS_STREAM_NEG, ctr) * (uint64_t)n_order) >> 32)];
    const int64_t p0 = ptr[u], n = ptr[u + 1] - p0;
    const int64_t vp = items[p0 + (int64_t)(((uint64_t)re_rng_u32(key, RS_STREAM_NEG, ctr + 1) * (uint64_t)n) >> 32)];
    int64_t vn = 0;
    const int64_t* sl = sorted_items + p0;
    for (int t = 0; t < RS_MAX_TRIES; ++t) {
        vn = (int64_t)(((uint64_t)re_rng_u32(key, RS_STREAM_NEG, ctr + 2 + t) * (uint64_t)N) >> 32);
        int64_t lo = 0, hi = n;
        while (lo < hi) {
            const int64_t mid = (lo + hi) >> 1;
            if (sl[mid] < vn) lo = mid + 1; else hi = mid;
        }
        if (lo >= n || sl[lo] != vn) break;
    }
    users[b] = u; pos[b] = vp; neg[b] = vn;
}

extern "C" int re_seq_train_sample(const int64_t* ptr, const int64_t* items, const int64_t* sorted_items, const int64_t* order, int64_t n_order,
                                   int64_t b0, int64_t B, int64_t S, int64_t N, uint32_t seed, uint32_t step, int64_t* users, int64_t* seq,
                                   int64_t* pos, int64_t* neg, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!ptr || !items || !sorted_items || !order || !seq || !pos || !neg || B < 0 || S < 1 || N < 1 || n_order < 0 || b0 < 0) return RE_EINVAL;
    if (B * S > (int64_t)1 << 26) return RE_EUNSUPPORTED;          // (the draw counter is 32 bits: position * RS_MAX_TRIES)
    hipLaunchKernelGGL(seq_train_sample_k, dim3((unsigned)((B * S + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ptr, items, sorted_items, order,
                       n_order, b0, (int)B, (int)S, N, seed, step, users, seq, pos, neg);
    return re_launch_status();
}

extern "C" int re_gen_train_sample(const int64_t* ptr, const int64_t* items, const int64_t* sorted_items, const int64_t* order, int64_t n_order,
                                   int64_t B, int64_t N, uint32_t seed, uint32_t step, int64_t* users, int64_t* pos, int64_t* neg,
                                   re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!ptr || !items || !sorted_items || !order || !users || !pos || !neg || B < 0 || N < 1 || n_order < 1) return RE_EINVAL;
    if (B > (int64_t)1 << 26) return RE_EUNSUPPORTED;
    hipLaunchKernelGGL(gen_train_sample_k, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, ptr, items, sorted_items, order,
                       n_order, (int)B, N, seed, step, users, pos, neg);
    return re_launch_status();
}
