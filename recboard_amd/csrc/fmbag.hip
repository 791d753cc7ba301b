// K9: DeepFM multi-field embedding bag + FM second-order term + logistic-regression term, and BCE-with-logits.
//
// Reference (DeepFM/main.py): per field f an nn.Embedding(count_f, D=10) and an nn.Embedding(count_f, 1) (:46-54,:140-149);
//   E[b,f,:] = emb_f[x[b,f]]                      (:204-206, 2F tiny gather kernels + cat in aten)
//   lr[b]    = sum_f emb_lr_f[x[b,f]] + bias      (:58-62)
//   fm[b]    = 0.5 * sum_d ((sum_f E)^2 - sum_f E^2)   (:80-85)
//   logit    = lr + fm + dnn(flatten E);  loss = BCELoss4Logits(mean)(logit, label)   (:208-215)
// Layout: the F tables are concatenated into ONE [sum count_f, D] table (and one [sum count_f] LR vector); a field's
// ids are offset by the field's first row, so the whole multi-table lookup is one kernel and its gradient ONE
// scatter-add.  Rows are 40 B (D=10): sub-cache-line, the tables (Frappe: 5.4 k rows = 216 KB) live in L2.
// One thread per (row b, field f) reads its 10-float row with 8-byte loads; the per-row sums over fields are
// reduced across the F lanes of a row group with shuffles (F <= 64).
//
// Algorithmic bytes per row: F * (8 + 4D + 4) gathered + 4*F*D written.
#include <math.h>

#include "re_common.h"

#define FB_MAXD 16

// lanes [0, F) of a group of FP lanes (power of two >= F) handle the F fields of one batch row
template <int FP>
__global__ __launch_bounds__(256) void fm_bag_fwd_k(const float* __restrict__ T, const float* __restrict__ TL,
                                                    const float* __restrict__ lr_bias, const int64_t* __restrict__ offsets,
                                                    int64_t rows_total, const int64_t* __restrict__ x, int64_t B, int F, int D,
                                                    float* __restrict__ E, float* __restrict__ fm_lr, int64_t* __restrict__ rows_out) {
    const int f = threadIdx.x % FP;
    const int64_t b = (int64_t)blockIdx.x * (256 / FP) + threadIdx.x / FP;
    float e[FB_MAXD];
    float lr = 0.f;
    const bool act = b < B && f < F;
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) e[d] = 0.f;
    if (act) {
        const int64_t r = x[b * F + f] + offsets[f];
        if (rows_out) rows_out[b * F + f] = r;
        if (r >= 0 && r < rows_total) {
            for (int d = 0; d < D; ++d) e[d] = T[r * D + d];
            lr = TL[r];
        }
        float* dst = E + (b * F + f) * D;
        for (int d = 0; d < D; ++d) dst[d] = e[d];
    }
    // fm = 0.5 * sum_d ((sum_f e)^2 - sum_f e^2)
    float fm = 0.f;
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) {
        float s = e[d], q = e[d] * e[d];
#pragma unroll
        for (int o = FP / 2; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        fm += s * s - q;
    }
#pragma unroll
    for (int o = FP / 2; o > 0; o >>= 1) lr += __shfl_xor(lr, o, 64);
    if (act && f == 0) fm_lr[b] = 0.5f * fm + lr + lr_bias[0];
}

// gE[b,f,:] = dE_mlp[b,f,:] + dlogit[b] * (sum_f' E[b,f',:] - E[b,f,:]);  gL[b,f] = dlogit[b]
template <int FP>
__global__ __launch_bounds__(256) void fm_bag_bwd_k(const float* __restrict__ E, const float* __restrict__ dE_mlp,
                                                    const float* __restrict__ dlogit, int64_t B, int F, int D,
                                                    float* __restrict__ gE, float* __restrict__ gL) {
    const int f = threadIdx.x % FP;
    const int64_t b = (int64_t)blockIdx.x * (256 / FP) + threadIdx.x / FP;
    const bool act = b < B && f < F;
    float e[FB_MAXD];
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) e[d] = 0.f;
    if (act)
        for (int d = 0; d < D; ++d) e[d] = E[(b * F + f) * D + d];
    const float dl = (b < B) ? dlogit[b] : 0.f;
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) {
        float s = e[d];
#pragma unroll
        for (int o = FP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (act && d < D) {
            const int64_t i = (b * F + f) * D + d;
            gE[i] = (dE_mlp ? dE_mlp[i] : 0.f) + dl * (s - e[d]);
        }
    }
    if (act) gL[b * F + f] = dl;
}

extern "C" int re_fm_bag_fwd(const float* T, const float* TL, const float* lr_bias, const int64_t* offsets, int64_t rows_total,
                             const int64_t* x, int64_t B, int64_t F, int64_t D, float* E, float* fm_lr, int64_t* rows_out, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!T || !TL || !lr_bias || !offsets || !x || !E || !fm_lr || B < 0 || rows_total <= 0) return RE_EINVAL;
    if (F < 1 || F > 64 || D < 1 || D > FB_MAXD) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define FB_FWD(FPV) hipLaunchKernelGGL(fm_bag_fwd_k<FPV>, dim3((unsigned)re_cdiv(B, 256 / FPV)), dim3(256), 0, s, T, TL, lr_bias, offsets, rows_total, x, B, (int)F, (int)D, E, fm_lr, rows_out)
    if (F <= 8) FB_FWD(8); else if (F <= 16) FB_FWD(16); else if (F <= 32) FB_FWD(32); else FB_FWD(64);
#undef FB_FWD
    return re_launch_status();
}

extern "C" int re_fm_bag_bwd(const float* E, const float* dE_mlp, const float* dlogit, int64_t B, int64_t F, int64_t D, float* gE,
                             float* gL, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!E || !dlogit || !gE || !gL || B < 0) return RE_EINVAL;
    if (F < 1 || F > 64 || D < 1 || D > FB_MAXD) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define FB_BWD(FPV) hipLaunchKernelGGL(fm_bag_bwd_k<FPV>, dim3((unsigned)re_cdiv(B, 256 / FPV)), dim3(256), 0, s, E, dE_mlp, dlogit, B, (int)F, (int)D, gE, gL)
    if (F <= 8) FB_BWD(8); else if (F <= 16) FB_BWD(16); else if (F <= 32) FB_BWD(32); else FB_BWD(64);
#undef FB_BWD
    return re_launch_status();
}

// BCE with logits, reduction = mean (DeepFM/main.py:214): loss = mean(max(x,0) - x*y + log1p(exp(-|x|)));
// dlogit = (sigmoid(x) - y) / n;  dsum[0] = sum(dlogit) (gradient of the LR bias).  One block, fixed-order sums.
__global__ __launch_bounds__(1024) void bce_logits_k(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                     float* __restrict__ loss, float* __restrict__ dlogit, float* __restrict__ dsum) {
    __shared__ float s1[16], s2[16];
    float l = 0.f, g = 0.f;
    const float inv = 1.0f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float xi = x[i], yi = y[i];
        l += fmaxf(xi, 0.f) - xi * yi + log1pf(expf(-fabsf(xi)));
        const float d = (re_sigmoid(xi) - yi) * inv;
        dlogit[i] = d;
        g += d;
    }
    l = re_wave_sum(l);
    g = re_wave_sum(g);
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = l; s2[threadIdx.x >> 6] = g; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; ++w) { a += s1[w]; b += s2[w]; }
        loss[0] = a * inv;
        if (dsum) dsum[0] = b;
    }
}

extern "C" int re_bce_logits(const float* logits, const float* labels, int64_t n, float* loss, float* dlogit, float* dsum,
                             re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !loss || !dlogit || n <= 0) return RE_EINVAL;
    hipLaunchKernelGGL(bce_logits_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, labels, n, loss, dlogit, dsum);
    return re_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Cross entropy over materialised logits, forward + backward in place (SASRec --loss CE, SASRec/main.py:217-219;
// CrossEntropy4Logits(reduction="mean") = F.cross_entropy).  One workgroup per row: max, sum-exp, then
//   row_loss[m] = logsumexp(x_m) - x_m[y_m];   x_m <- (softmax(x_m) - onehot(y_m)) * (1 / M)      (the gradient)
// loss[0] = mean(row_loss) by a second, fixed-order kernel.  [M, N] = 3 000 x 12 101 on the benchmark shapes: 145 MB,
// read twice and written once.
__global__ __launch_bounds__(256) void ce_rows_k(float* __restrict__ logits, int64_t N, int64_t ld, const int64_t* __restrict__ labels,
                                                 float inv_m, float* __restrict__ row_loss) {
    __shared__ float red[4];
    float* x = logits + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int64_t i = tid; i < N; i += 256) mx = fmaxf(mx, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int64_t i = tid; i < N; i += 256) s += expf(x[i] - mx);
    s = re_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = ((red[0] + red[1]) + red[2]) + red[3];
    const int64_t y = labels[blockIdx.x];
    const bool yok = y >= 0 && y < N;
    if (tid == 0) row_loss[blockIdx.x] = yok ? (logf(s) + mx - x[y]) : 0.f;
    __syncthreads();
    const float inv_s = 1.0f / s;
    for (int64_t i = tid; i < N; i += 256) {
        float p = expf(x[i] - mx) * inv_s;
        if (yok && i == y) p -= 1.0f;
        x[i] = p * inv_m;
    }
}

__global__ __launch_bounds__(256) void sum_mean_k(const float* __restrict__ v, int64_t n, float scale, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
    s = re_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (((red[0] + red[1]) + red[2]) + red[3]) * scale;
}

extern "C" int re_ce_rows(float* logits, int64_t M, int64_t N, int64_t ld, const int64_t* labels, float* row_loss, float* loss,
                          re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !row_loss || !loss || M <= 0 || N <= 0 || ld < N) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_rows_k, dim3((unsigned)M), dim3(256), 0, s, logits, N, ld, labels, 1.0f / (float)M, row_loss);
    hipLaunchKernelGGL(sum_mean_k, dim3(1), dim3(256), 0, s, (const float*)row_loss, M, 1.0f / (float)M, loss);
    return re_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// The same cross entropy WITHOUT the [M, N] matrix: the catalog is walked in column chunks, the caller materialises one chunk of
// logits [M, Nc] at a time (one GEMM), and the row statistics are carried between chunks (online log-sum-exp):
//   pass 1, per chunk:  (rowmax, rowsum) <- merge with the chunk's (max, sum exp(x - max)); tgt[m] = x[m, y_m] when y_m is in the chunk
//   finish:             row_loss = log(rowsum) + rowmax - tgt;  loss = mean
//   pass 2, per chunk (logits recomputed by the same GEMM):  x <- (exp(x - rowmax) / rowsum - [col == y_m]) / M   in place
// Memory is M x Nc instead of M x N (a 1 M-item catalog at M = 4 000 would be 16 GB of logits); the price is one more GEMM per chunk.
__global__ __launch_bounds__(256) void ce_chunk_stats_k(const float* __restrict__ logits, int64_t Nc, int64_t ld, int64_t col0,
                                                        const int64_t* __restrict__ labels, int first, float* __restrict__ rowmax,
                                                        float* __restrict__ rowsum, float* __restrict__ tgt) {
    __shared__ float red[4];
    const float* x = logits + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int64_t i = tid; i < Nc; i += 256) mx = fmaxf(mx, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int64_t i = tid; i < Nc; i += 256) s += expf(x[i] - mx);
    s = re_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        s = ((red[0] + red[1]) + red[2]) + red[3];
        if (!first) {
            const float m0 = rowmax[blockIdx.x], s0 = rowsum[blockIdx.x];
            const float m1 = fmaxf(m0, mx);
            s = s0 * expf(m0 - m1) + s * expf(mx - m1);
            mx = m1;
        }
        rowmax[blockIdx.x] = mx;
        rowsum[blockIdx.x] = s;
        const int64_t y = labels[blockIdx.x] - col0;
        if (y >= 0 && y < Nc) tgt[blockIdx.x] = x[y];
        else if (first) tgt[blockIdx.x] = 0.f;
    }
}

__global__ __launch_bounds__(256) void ce_chunk_loss_k(const float* __restrict__ rowmax, const float* __restrict__ rowsum, const float* __restrict__ tgt,
                                                       const int64_t* __restrict__ labels, int64_t M, int64_t N, float* __restrict__ row_loss) {
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
        const int64_t y = labels[m];
        row_loss[m] = (y >= 0 && y < N) ? logf(rowsum[m]) + rowmax[m] - tgt[m] : 0.f;
    }
}

__global__ __launch_bounds__(256) void ce_chunk_grad_k(float* __restrict__ logits, int64_t Nc, int64_t ld, int64_t col0,
                                                       const int64_t* __restrict__ labels, const float* __restrict__ rowmax,
                                                       const float* __restrict__ rowsum, float inv_m) {
    float* x = logits + (int64_t)blockIdx.x * ld;
    const float mx = rowmax[blockIdx.x], inv_s = 1.0f / rowsum[blockIdx.x];
    const int64_t y = labels[blockIdx.x] - col0;
    for (int64_t i = threadIdx.x; i < Nc; i += 256) {
        float p = expf(x[i] - mx) * inv_s;
        if (i == y) p -= 1.0f;
        x[i] = p * inv_m;
    }
}

extern "C" int re_ce_chunk_stats(const float* logits, int64_t M, int64_t Nc, int64_t ld, int64_t col0, const int64_t* labels, int first,
                                 float* rowmax, float* rowsum, float* tgt, re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !rowmax || !rowsum || !tgt || M <= 0 || Nc <= 0 || ld < Nc || col0 < 0) return RE_EINVAL;
    hipLaunchKernelGGL(ce_chunk_stats_k, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, logits, Nc, ld, col0, labels, first, rowmax, rowsum, tgt);
    return re_launch_status();
}

extern "C" int re_ce_chunk_loss(const float* rowmax, const float* rowsum, const float* tgt, const int64_t* labels, int64_t M, int64_t N,
                                float* row_loss, float* loss, re_stream_t stream) {
    re_clear_error();
    if (!rowmax || !rowsum || !tgt || !labels || !row_loss || !loss || M <= 0 || N <= 0) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_chunk_loss_k, dim3(re_grid(M, 256)), dim3(256), 0, s, rowmax, rowsum, tgt, labels, M, N, row_loss);
    hipLaunchKernelGGL(sum_mean_k, dim3(1), dim3(256), 0, s, (const float*)row_loss, M, 1.0f / (float)M, loss);
    return re_launch_status();
}

extern "C" int re_ce_chunk_grad(float* logits, int64_t M, int64_t Nc, int64_t ld, int64_t col0, const int64_t* labels, const float* rowmax,
                                const float* rowsum, re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !rowmax || !rowsum || M <= 0 || Nc <= 0 || ld < Nc || col0 < 0) return RE_EINVAL;
    hipLaunchKernelGGL(ce_chunk_grad_k, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, logits, Nc, ld, col0, labels, rowmax, rowsum,
                       1.0f / (float)M);
    return re_launch_status();
}
