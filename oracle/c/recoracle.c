/* CPU oracle, C part -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * Plain-C restatement of the integer/ordering-sensitive pieces of the hot path, used by tests/ and by
 * bench.py's cpu_baseline leg only:
 *
 *   ro_score_dense   scores[b][n] = sum_k Q[b][k] * E[n][k]   -- SASRec/main.py:228 einsum("BD,ND->BN"),
 *                    MF-BPR/main.py:104, LightGCN/main.py:120.  Accumulated as a k-ordered fp32 `fmaf`
 *                    chain starting from 0: bit-for-bit what gfx950's v_mfma_f32_32x32x2_f32 computes
 *                    (cdna_hip_programming.md §3 "FP32-input MFMA": D = fma(a_k1,b_k1, fma(a_k0,b_k0, C))).
 *   ro_score_topk    Coach.evaluate full-ranking contract, mirrored at UniSRec/main.py:408-414:
 *                    scores[seen] = -1e23, then top-K (sorted descending).  torch.topk leaves tie order
 *                    unspecified; the engine DEFINES ties -> lowest item index, and so does this oracle.
 *   ro_gather_rows / ro_scatter_add_rows   embedding lookup and its dense gradient, position order.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no implicit fusing, fmaf only where written).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define RO_MASKED (-1e23f)

void ro_score_dense(const float* Q, const float* E, int64_t B, int64_t N, int64_t D, float* out) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t n = 0; n < N; ++n) {
            float acc = 0.0f;
            const float* q = Q + b * D;
            const float* e = E + n * D;
            for (int64_t k = 0; k < D; ++k) acc = fmaf(q[k], e[k], acc);
            out[b * N + n] = acc;
        }
}

/* a ranks before b  <=>  higher value, or equal value and lower index */
static inline int ro_before(float va, int64_t ia, float vb, int64_t ib) {
    return (va > vb) || (va == vb && ia < ib);
}

/* seen_ptr[B+1], seen_idx[nnz] (any order, duplicates allowed); seen_ptr may be NULL (= retain_seen). */
void ro_score_topk(const float* Q, const float* E, int64_t B, int64_t N, int64_t D,
                   const int64_t* seen_ptr, const int64_t* seen_idx, int64_t K,
                   float* vals, int64_t* idx) {
    float* row = (float*)malloc(sizeof(float) * (size_t)N);
    for (int64_t b = 0; b < B; ++b) {
        ro_score_dense(Q + b * D, E, 1, N, D, row);
        if (seen_ptr)
            for (int64_t p = seen_ptr[b]; p < seen_ptr[b + 1]; ++p)
                if (seen_idx[p] >= 0 && seen_idx[p] < N) row[seen_idx[p]] = RO_MASKED;
        /* insertion into a sorted list of K: O(N*K) worst case, fine for oracle sizes */
        float* v = vals + b * K;
        int64_t* ix = idx + b * K;
        int64_t cnt = 0;
        for (int64_t n = 0; n < N; ++n) {
            float s = row[n];
            if (cnt == K && !ro_before(s, n, v[K - 1], ix[K - 1])) continue;
            int64_t j = cnt < K ? cnt : K - 1;
            while (j > 0 && ro_before(s, n, v[j - 1], ix[j - 1])) { v[j] = v[j - 1]; ix[j] = ix[j - 1]; --j; }
            v[j] = s; ix[j] = n;
            if (cnt < K) ++cnt;
        }
        for (int64_t j = cnt; j < K; ++j) { v[j] = -INFINITY; ix[j] = -1; }  /* K > N */
    }
    free(row);
}

void ro_gather_rows(const float* W, const int64_t* idx, int64_t n, int64_t D, float* out) {
    for (int64_t i = 0; i < n; ++i) memcpy(out + i * D, W + idx[i] * D, sizeof(float) * (size_t)D);
}

/* dense [R,D] gradient; contributions added in position order; rows == padding_idx skipped */
void ro_scatter_add_rows(const float* g, const int64_t* idx, int64_t n, int64_t D, int64_t R,
                         int64_t padding_idx, float* out) {
    memset(out, 0, sizeof(float) * (size_t)(R * D));
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = idx[i];
        if (r == padding_idx) continue;
        for (int64_t d = 0; d < D; ++d) out[r * D + d] += g[i * D + d];
    }
}
