// DeepFM MLP pieces around the GEMM: BatchNorm1d (training: batch statistics; eval: running statistics) + ReLU + Dropout,
// forward and backward, and column sums for bias gradients.                                   HBM-bound, [M, N] = 4096 x 400.
//
// Reference: MLPBlock.forward = dropout(relu(bn(linear(x))))  (DeepFM/main.py:119-124), nn.BatchNorm1d defaults
// (eps 1e-5, momentum 0.1, biased variance for normalisation, unbiased for the running estimate).
// Column statistics: a workgroup owns 64 columns; thread (c, rg) accumulates rows rg, rg+4, ... (coalesced over c), the
// four row groups are combined in LDS in a fixed order -> deterministic.  Variance is two-pass (mean first).
#include <math.h>

#include "re_common.h"
#include "re_rng.h"

// Column reductions over [M, N] run on a (N/64) x ML_CHUNKS grid: block (bx, by) reduces rows [by*rpc, (by+1)*rpc) of 64
// columns (thread (c, rg) takes rows rg, rg+4, ... of the chunk: coalesced over c), the four row groups are combined in
// LDS in a fixed order, and a finalize kernel merges the ML_CHUNKS partials per column in chunk order -> deterministic.
#define ML_CHUNKS 64

struct MlNoPost { __device__ __forceinline__ void operator()(int64_t, int64_t, float) const {} };
template <class F, class P = MlNoPost>
__device__ __forceinline__ void col_reduce2(int64_t M, int64_t N, F f, float& o1, float& o2, bool& owner, int64_t& col,
                                            int64_t& m_begin, int64_t& m_end, P post = P{}) {
    __shared__ float r1[256], r2[256];
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    col = (int64_t)blockIdx.x * 64 + c;
    const int64_t rpc = (M + gridDim.y - 1) / gridDim.y;
    m_begin = (int64_t)blockIdx.y * rpc;
    m_end = (m_begin + rpc < M) ? m_begin + rpc : M;
    // (four rows a trip, their loads independent of each other -- one row a trip is a memory round trip per row: 29 us for the 13 MB of a
    //  [4096, 400] backward pass -- added in row order)
    float a = 0.f, b = 0.f;
    if (col < N) {
        int64_t m = m_begin + rg;
        for (; m + 28 < m_end; m += 32) {     // (eight rows a trip)
            float x[8], y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) f(m + 4 * i, col, x[i], y[i]);
#pragma unroll
            for (int i = 0; i < 8; ++i) post(m + 4 * i, col, x[i]);      // (a caller's stores: behind the group's loads)
#pragma unroll
            for (int i = 0; i < 8; ++i) { a += x[i]; b += y[i]; }
        }
        for (; m + 12 < m_end; m += 16) {
            float x0, y0, x1, y1, x2, y2, x3, y3;
            f(m, col, x0, y0); f(m + 4, col, x1, y1); f(m + 8, col, x2, y2); f(m + 12, col, x3, y3);
            post(m, col, x0); post(m + 4, col, x1); post(m + 8, col, x2); post(m + 12, col, x3);
            a += x0; b += y0; a += x1; b += y1; a += x2; b += y2; a += x3; b += y3;
        }
        for (; m < m_end; m += 4) { float x, y; f(m, col, x, y); post(m, col, x); a += x; b += y; }
    }
    r1[threadIdx.x] = a; r2[threadIdx.x] = b;
    __syncthreads();
    owner = rg == 0 && col < N;
    if (owner) {
        o1 = ((r1[c] + r1[64 + c]) + r1[128 + c]) + r1[192 + c];
        o2 = ((r2[c] + r2[64 + c]) + r2[128 + c]) + r2[192 + c];
    }
    __syncthreads();
}

// partial[by][0][n] = chunk mean, partial[by][1][n] = chunk M2 (sum of squared deviations from the chunk mean)
__global__ __launch_bounds__(256) void bn_stats_partial_k(const float* __restrict__ z, int64_t M, int64_t N, float* __restrict__ partial) {
    float s, dummy, q;
    bool owner;
    int64_t col, mb, me;
    __shared__ float s_mean[64];
    col_reduce2(M, N, [&](int64_t m, int64_t n, float& x, float& y) { x = z[m * N + n]; y = 0.f; }, s, dummy, owner, col, mb, me);
    const float cnt = (float)(me > mb ? me - mb : 0);
    if (owner) s_mean[threadIdx.x & 63] = cnt > 0.f ? s / cnt : 0.f;
    __syncthreads();
    const float mu = s_mean[threadIdx.x & 63];
    col_reduce2(M, N, [&](int64_t m, int64_t n, float& x, float& y) { const float d = z[m * N + n] - mu; x = d * d; y = 0.f; }, q, dummy, owner, col, mb, me);
    if (owner) {
        partial[((int64_t)blockIdx.y * 2 + 0) * N + col] = mu;
        partial[((int64_t)blockIdx.y * 2 + 1) * N + col] = q;
    }
}

// Chan's parallel-variance merge of the chunk (mean, M2, count) triples; then (mean, rstd) + running statistics.  One WAVE per column: lane b
// holds chunk b's triple and the 64 lanes merge in a fixed butterfly (lane ^ 1, ^ 2, ... ^ 32: the same tree for every column and every run) --
// a thread per column walking the chunks one after the other is a chain of 64 dependent merges with two divisions each (12 - 14 us for 400
// columns on two workgroups; 0.83 ms step: DeepFM's three BatchNorms call it every step).
static_assert(ML_CHUNKS <= 64, "one lane per chunk");
__global__ __launch_bounds__(256) void bn_stats_final_k(const float* __restrict__ partial, int chunks, int64_t M, int64_t N, float eps,
                                                        float momentum, float* __restrict__ stats, float* __restrict__ run_mean,
                                                        float* __restrict__ run_var) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;                                   // (whole waves)
    const int64_t rpc = (M + chunks - 1) / chunks;
    const int64_t mb = (int64_t)lane * rpc;
    const bool live = lane < chunks && mb < M;
    float cnt = live ? (float)(((mb + rpc < M) ? mb + rpc : M) - mb) : 0.f;
    const int bb = live ? lane : 0;
    float mean = partial[((int64_t)bb * 2 + 0) * N + n], m2 = partial[((int64_t)bb * 2 + 1) * N + n];
    if (!live) { mean = 0.f; m2 = 0.f; }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float mo = __shfl_xor(mean, o, 64), qo = __shfl_xor(m2, o, 64), co = __shfl_xor(cnt, o, 64);
        // the pair (lower lane's triple, upper lane's triple), merged the same way in both lanes
        const bool up = (lane & o) != 0;
        const float ma = up ? mo : mean, qa = up ? qo : m2, ca = up ? co : cnt;
        const float mb_ = up ? mean : mo, qb = up ? m2 : qo, cb = up ? cnt : co;
        const float tot = ca + cb;
        const float w = tot > 0.f ? cb / tot : 0.f;
        const float delta = mb_ - ma;
        mean = ma + delta * w;
        m2 = qa + qb + delta * delta * (ca * w);
        cnt = tot;
    }
    if (lane != 0) return;
    const float var = m2 / (float)M;
    stats[n] = mean;
    stats[N + n] = 1.0f / sqrtf(var + eps);
    if (run_mean) {
        run_mean[n] = (1.f - momentum) * run_mean[n] + momentum * mean;
        run_var[n] = (1.f - momentum) * run_var[n] + momentum * (M > 1 ? m2 / (float)(M - 1) : var);
    }
}

// out1[n] = sum over chunks of partial[b][0][n], out2 likewise (either may be null)
__global__ __launch_bounds__(256) void col_final_k(const float* __restrict__ partial, int chunks, int64_t N, float* __restrict__ out1,
                                                   float* __restrict__ out2) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float a = 0.f, b = 0.f;
    float pa[ML_CHUNKS], pb[ML_CHUNKS];      // (requested together, added in chunk order)
#pragma unroll
    for (int c = 0; c < ML_CHUNKS; ++c) {
        const int cc = c < chunks ? c : chunks - 1;
        pa[c] = partial[((int64_t)cc * 2 + 0) * N + n];
        pb[c] = partial[((int64_t)cc * 2 + 1) * N + n];
    }
#pragma unroll
    for (int c = 0; c < ML_CHUNKS; ++c)
        if (c < chunks) { a += pa[c]; b += pb[c]; }
    if (out1) out1[n] = a;
    if (out2) out2[n] = b;
}

// eval mode: stats from the running estimates
__global__ __launch_bounds__(256) void bn_stats_eval_k(const float* __restrict__ run_mean, const float* __restrict__ run_var, int64_t N,
                                                       float eps, float* __restrict__ stats) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n < N) { stats[n] = run_mean[n]; stats[N + n] = 1.0f / sqrtf(run_var[n] + eps); }
}

// a = dropout(relu((z - mean) * rstd * gamma + beta));   stats == NULL: no BatchNorm (a = dropout(relu(z)))
__global__ __launch_bounds__(256) void bn_relu_drop_fwd_k(const float* __restrict__ z, int64_t total, int64_t N,
                                                          const float* __restrict__ stats, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float drop_scale, uint32_t thresh,
                                                          uint32_t seed, uint32_t stream_id, float* __restrict__ a,
                                                          const uint32_t* __restrict__ seed_dev) {
    if (seed_dev) seed = seed_dev[0];   // (captured steps: the step's seed is a device word)
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e % N;
        float v = z[e];
        if (stats) v = (v - stats[n]) * stats[N + n] * gamma[n] + beta[n];
        v = fmaxf(v, 0.f);
        if (thresh) v = re_keep(seed, stream_id, (uint32_t)e, thresh) ? v * drop_scale : 0.f;
        a[e] = v;
    }
}

// backward, pass 1: g = da * dropmask * (a > 0)  (written to dz as scratch); partial sums of g and g*xhat per chunk
__global__ __launch_bounds__(256) void bn_bwd_reduce_k(const float* __restrict__ da, const float* __restrict__ a, const float* __restrict__ z,
                                                       int64_t M, int64_t N, const float* __restrict__ stats, float drop_scale,
                                                       float* __restrict__ dz, float* __restrict__ partial) {
    float sg, sgx;
    bool owner;
    int64_t col, mb, me;
    // (the thread's column statistics: read once -- behind the stores to dz the compiler has to assume they changed)
    const int64_t myc = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const bool bn = stats != nullptr;
    const float mu = (bn && myc < N) ? stats[myc] : 0.f, rs = (bn && myc < N) ? stats[N + myc] : 0.f;
    col_reduce2(M, N, [&](int64_t m, int64_t n, float& x, float& y) {
        const int64_t e = m * N + n;
        const float av = a[e], dv = da[e], zv = z[e];              // (unconditional loads: a load behind a compare is a round trip of its own)
        const float g = (av > 0.f) ? dv * drop_scale : 0.f;        // a > 0 <=> relu active and not dropped
        x = g;
        y = bn ? g * (zv - mu) * rs : 0.f;
    }, sg, sgx, owner, col, mb, me, [&](int64_t m, int64_t n, float g) { dz[m * N + n] = g; });
    if (owner) {
        partial[((int64_t)blockIdx.y * 2 + 0) * N + col] = sg;
        partial[((int64_t)blockIdx.y * 2 + 1) * N + col] = sgx;
    }
}

// backward, pass 2 (BatchNorm only): dz = rstd * gamma * (g - mean(g) - xhat * mean(g * xhat))
__global__ __launch_bounds__(256) void bn_bwd_apply_k(float* __restrict__ dz, const float* __restrict__ z, int64_t total, int64_t M, int64_t N,
                                                      const float* __restrict__ stats, const float* __restrict__ gamma,
                                                      const float* __restrict__ dgamma, const float* __restrict__ dbeta) {
    const float invm = 1.0f / (float)M;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t n = e % N;
        const float xhat = (z[e] - stats[n]) * stats[N + n];
        dz[e] = stats[N + n] * gamma[n] * (dz[e] - dbeta[n] * invm - xhat * dgamma[n] * invm);
    }
}

__global__ __launch_bounds__(256) void colsum_k(const float* __restrict__ x, int64_t M, int64_t N, float* __restrict__ partial) {
    float s, d;
    bool owner;
    int64_t col, mb, me;
    col_reduce2(M, N, [&](int64_t m, int64_t n, float& a, float& b) { a = x[m * N + n]; b = 0.f; }, s, d, owner, col, mb, me);
    if (owner) {
        partial[((int64_t)blockIdx.y * 2 + 0) * N + col] = s;
        partial[((int64_t)blockIdx.y * 2 + 1) * N + col] = 0.f;
    }
}


// ---- one-launch forms (the producing GEMM's epilogue left per-chunk partials: re_gemm_f32_colstats / re_gemm_f32_gated)
// A workgroup = 64 columns x one block of rows.  It first reduces the chunk partials of ITS columns (every row block of a column repeats the
// same arithmetic on the same numbers: identical results, no second launch, no flag), then streams its rows.
static int ml_row_blocks(int64_t M) { const int64_t b = re_cdiv(M, 64); return (int)(b < 1 ? 1 : (b > 128 ? 128 : b)); }

// chunk b of `colstats` = (mean, M2) over rows [b rpc, (b + 1) rpc), rpc = ceil(M / chunks).  Thread (c, q) merges chunks [q per, (q + 1) per) one
// after the other (Chan), the four quarter results are merged ((0, 1), (2, 3)).
__global__ __launch_bounds__(256) void bn_fwd_fused_k(const float* __restrict__ z, int64_t M, int64_t N, const float* __restrict__ colstats, int chunks,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta, float eps, float momentum,
                                                      float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ stats,
                                                      float drop_scale, uint32_t thresh, uint32_t seed, uint32_t stream_id, float* __restrict__ a,
                                                      const uint32_t* __restrict__ seed_dev) {
    __shared__ float s_mean[256], s_m2[256], s_cnt[256];
    if (seed_dev) seed = seed_dev[0];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const bool ok = col < N;
    const int64_t rpc = (M + chunks - 1) / chunks;
    const int per = (chunks + 3) >> 2;
    float mean = 0.f, m2 = 0.f, cnt = 0.f;
    if (ok) {
        const int lo = q * per, hi = (lo + per < chunks) ? lo + per : chunks;
        // (sixteen chunks' partials requested together, through clamped indices: one at a time behind a `break` they were sixteen dependent
        //  L2 round trips -- 8 of the launch's 11 us at M = 4 096; the merge itself is the same chain of Chan updates, in chunk order)
        for (int b0 = lo; b0 < hi; b0 += 16) {
            float mo[16], qo[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = b0 + i < hi ? b0 + i : hi - 1;
                mo[i] = colstats[((int64_t)b * 2 + 0) * N + col];
                qo[i] = colstats[((int64_t)b * 2 + 1) * N + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int64_t mb = (int64_t)(b0 + i) * rpc;
                if (b0 + i < hi && mb < M) {
                    const float cb = (float)(((mb + rpc < M) ? mb + rpc : M) - mb);
                    const float tot = cnt + cb, w = cb / tot, delta = mo[i] - mean;
                    mean = fmaf(delta, w, mean);
                    m2 = m2 + qo[i] + delta * delta * (cnt * w);
                    cnt = tot;
                }
            }
        }
    }
    s_mean[threadIdx.x] = mean; s_m2[threadIdx.x] = m2; s_cnt[threadIdx.x] = cnt;
    __syncthreads();
    {
        auto merge = [](float ma, float qa, float ca, float mb_, float qb, float cb, float& mo, float& qo, float& co) {
            const float tot = ca + cb, w = tot > 0.f ? cb / tot : 0.f, delta = mb_ - ma;
            mo = fmaf(delta, w, ma); qo = qa + qb + delta * delta * (ca * w); co = tot;
        };
        float m01, q01, c01, m23, q23, c23;
        merge(s_mean[c], s_m2[c], s_cnt[c], s_mean[64 + c], s_m2[64 + c], s_cnt[64 + c], m01, q01, c01);
        merge(s_mean[128 + c], s_m2[128 + c], s_cnt[128 + c], s_mean[192 + c], s_m2[192 + c], s_cnt[192 + c], m23, q23, c23);
        merge(m01, q01, c01, m23, q23, c23, mean, m2, cnt);
    }
    if (!ok) return;
    const float var = m2 / (float)M;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (blockIdx.y == 0 && q == 0) {
        stats[col] = mean;
        stats[N + col] = rstd;
        if (run_mean) {
            run_mean[col] = (1.f - momentum) * run_mean[col] + momentum * mean;
            run_var[col] = (1.f - momentum) * run_var[col] + momentum * (M > 1 ? m2 / (float)(M - 1) : var);
        }
    }
    const float k1 = gamma[col], k0 = beta[col];
    const int64_t rpb = (M + gridDim.y - 1) / gridDim.y;
    const int64_t m_begin = (int64_t)blockIdx.y * rpb, m_end = (m_begin + rpb < M) ? m_begin + rpb : M;
    auto one = [&](int64_t e, float v) {
        v = (v - mean) * rstd * k1 + k0;                            // (bn_relu_drop_fwd_k's expression)
        v = fmaxf(v, 0.f);
        if (thresh) v = re_keep(seed, stream_id, (uint32_t)e, thresh) ? v * drop_scale : 0.f;
        a[e] = v;
    };
    int64_t m = m_begin + q;
    for (; m + 28 < m_end; m += 32) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = z[(m + 4 * i) * N + col];
#pragma unroll
        for (int i = 0; i < 8; ++i) one((m + 4 * i) * N + col, v[i]);
    }
    for (; m + 12 < m_end; m += 16) {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = z[(m + 4 * i) * N + col];
#pragma unroll
        for (int i = 0; i < 4; ++i) one((m + 4 * i) * N + col, v[i]);
    }
    for (; m < m_end; m += 4) one(m * N + col, z[m * N + col]);
}

// part [chunks][ps][N]: thread (c, q) adds chunks [q per, (q + 1) per) in order, the quarters are added ((0 + 1) + 2) + 3
__global__ __launch_bounds__(256) void bn_bwd_apply2_k(float* __restrict__ dz, const float* __restrict__ z, int64_t M, int64_t N,
                                                       const float* __restrict__ stats, const float* __restrict__ gamma,
                                                       const float* __restrict__ part, int chunks, int ps, float* __restrict__ dgamma,
                                                       float* __restrict__ dbeta, float* __restrict__ extra_out) {
    __shared__ float r0[256], r1[256], r2[256];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const bool ok = col < N;
    const bool third = extra_out != nullptr && blockIdx.y == 0;      // (uniform)
    const int per = (chunks + 3) >> 2;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (ok) {
        const int lo = q * per, hi = (lo + per < chunks) ? lo + per : chunks;
        // (sixteen chunks' sums requested together through clamped indices; added in chunk order: x + 0 = x past the end)
        for (int b0 = lo; b0 < hi; b0 += 16) {
            float p0[16], p1[16], p2[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int b = b0 + i < hi ? b0 + i : hi - 1;
                p0[i] = part[((int64_t)b * ps + 0) * N + col];
                p1[i] = part[((int64_t)b * ps + 1) * N + col];
                p2[i] = part[((int64_t)b * ps + (ps > 2 ? 2 : 1)) * N + col];
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool in = b0 + i < hi;
                s0 += in ? p0[i] : 0.f;
                s1 += in ? p1[i] : 0.f;
                s2 += (in && third) ? p2[i] : 0.f;
            }
        }
    }
    r0[threadIdx.x] = s0; r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    if (!ok) return;
    const float db = ((r0[c] + r0[64 + c]) + r0[128 + c]) + r0[192 + c];
    const float dg = ((r1[c] + r1[64 + c]) + r1[128 + c]) + r1[192 + c];
    if (blockIdx.y == 0 && q == 0) {
        dbeta[col] = db;
        dgamma[col] = dg;
        if (third) extra_out[col] = ((r2[c] + r2[64 + c]) + r2[128 + c]) + r2[192 + c];
    }
    const float invm = 1.0f / (float)M;
    const float mu = stats[col], rs = stats[N + col], k = rs * gamma[col];
    const int64_t rpb = (M + gridDim.y - 1) / gridDim.y;
    const int64_t m_begin = (int64_t)blockIdx.y * rpb, m_end = (m_begin + rpb < M) ? m_begin + rpb : M;
    auto one = [&](int64_t e, float gv, float zv) {
        const float xhat = (zv - mu) * rs;
        dz[e] = k * (gv - db * invm - xhat * dg * invm);             // (bn_bwd_apply_k's expression)
    };
    int64_t m = m_begin + q;
    for (; m + 28 < m_end; m += 32) {
        float gv[8], zv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { gv[i] = dz[(m + 4 * i) * N + col]; zv[i] = z[(m + 4 * i) * N + col]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) one((m + 4 * i) * N + col, gv[i], zv[i]);
    }
    for (; m + 12 < m_end; m += 16) {
        float gv[4], zv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { gv[i] = dz[(m + 4 * i) * N + col]; zv[i] = z[(m + 4 * i) * N + col]; }
#pragma unroll
        for (int i = 0; i < 4; ++i) one((m + 4 * i) * N + col, gv[i], zv[i]);
    }
    for (; m < m_end; m += 4) one(m * N + col, dz[m * N + col], z[m * N + col]);
}

static int ml_chunks(int64_t M) { return M >= 4 * ML_CHUNKS ? ML_CHUNKS : 1; }
extern "C" size_t re_mlp_workspace_bytes(int64_t N) { return (size_t)ML_CHUNKS * 2 * N * sizeof(float) + 256; }

extern "C" int re_bn_relu_drop_fwd(const float* z, int64_t M, int64_t N, const float* gamma, const float* beta, float* run_mean,
                                   float* run_var, int training, float eps, float momentum, float drop_p, uint32_t seed,
                                   const uint32_t* seed_dev, uint32_t stream_id, float* stats, float* a, void* ws, size_t ws_bytes,
                                   re_stream_t stream) {
    re_clear_error();
    if (!z || !a || M <= 0 || N <= 0) return RE_EINVAL;
    if (gamma && training && (!ws || ws_bytes < re_mlp_workspace_bytes(N))) return RE_EWORKSPACE;
    const bool bn = gamma != nullptr;
    if (bn && (!beta || !stats || !run_mean || !run_var)) return RE_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (bn) {
        if (training) {
            const int ch = ml_chunks(M);
            hipLaunchKernelGGL(bn_stats_partial_k, dim3((unsigned)re_cdiv(N, 64), ch), dim3(256), 0, s, z, M, N, (float*)ws);
            hipLaunchKernelGGL(bn_stats_final_k, dim3((unsigned)re_cdiv(N, 4)), dim3(256), 0, s, (const float*)ws, ch, M, N, eps, momentum, stats, run_mean, run_var);
        }
        else hipLaunchKernelGGL(bn_stats_eval_k, dim3((unsigned)re_cdiv(N, 256)), dim3(256), 0, s, (const float*)run_mean, (const float*)run_var, N, eps, stats);
    }
    const uint32_t thresh = (training && drop_p > 0.f) ? re_drop_threshold(drop_p) : 0u;
    const float ds = thresh ? 1.0f / (1.0f - drop_p) : 1.0f;
    hipLaunchKernelGGL(bn_relu_drop_fwd_k, dim3(re_grid(M * N, 1024)), dim3(256), 0, s, z, M * N, N, bn ? (const float*)stats : nullptr, gamma, beta, ds,
                       thresh, seed, stream_id, a, seed_dev);
    return re_launch_status();
}

// re_bn_relu_drop_fwd in training mode with the per-chunk (mean, M2) partials of z's columns ALREADY in `colstats` ([chunks][2][N], chunk b =
// rows [b rpc, (b + 1) rpc) with rpc = ceil(M / chunks): what re_gemm_f32_colstats leaves, chunks = M / 64): the pass over z that takes the
// statistics is the producing GEMM's epilogue; here the merge (+ running statistics) and the normalise / ReLU / dropout pass.
extern "C" int re_bn_relu_drop_fwd_pre(const float* z, int64_t M, int64_t N, const float* gamma, const float* beta, float* run_mean,
                                       float* run_var, float eps, float momentum, float drop_p, uint32_t seed, const uint32_t* seed_dev,
                                       uint32_t stream_id, float* stats, float* a, const float* colstats, int chunks, re_stream_t stream) {
    re_clear_error();
    if (!z || !a || !gamma || !beta || !stats || !run_mean || !run_var || !colstats || M <= 0 || N <= 0) return RE_EINVAL;
    if (chunks < 1 || drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = thresh ? 1.0f / (1.0f - drop_p) : 1.0f;
    hipLaunchKernelGGL(bn_fwd_fused_k, dim3((unsigned)re_cdiv(N, 64), (unsigned)ml_row_blocks(M)), dim3(256), 0, s, z, M, N, colstats, chunks, gamma, beta,
                       eps, momentum, run_mean, run_var, stats, ds, thresh, seed, stream_id, a, seed_dev);
    return re_launch_status();
}

// The second half of the backward of dropout(relu(bn(z))) when the gate and its column sums came out of the producing launch
// (re_gemm_f32_gated, re_mlp_head_bwd_gated): g [M, N] in, dz out in place; part [chunks][pstride][N] holds per-row-chunk (sum g, sum g xhat
// [, a third column sum]); dbeta = sum g, dgamma = sum g xhat (chunks added in a fixed order), dz = rstd gamma (g - dbeta / M - xhat dgamma / M);
// extra_out [N] (pstride == 3) = the third sums.  One launch: every workgroup adds the chunks of its 64 columns itself.
extern "C" int re_bn_bwd_apply(float* g, const float* z, int64_t M, int64_t N, const float* gamma, const float* stats, const float* part,
                               int chunks, int pstride, float* dgamma, float* dbeta, float* extra_out, re_stream_t stream) {
    re_clear_error();
    if (!g || !z || !gamma || !stats || !part || !dgamma || !dbeta || M <= 0 || N <= 0 || chunks < 1) return RE_EINVAL;
    if ((pstride != 2 && pstride != 3) || (extra_out && pstride != 3)) return RE_EINVAL;
    hipLaunchKernelGGL(bn_bwd_apply2_k, dim3((unsigned)re_cdiv(N, 64), (unsigned)ml_row_blocks(M)), dim3(256), 0, (hipStream_t)stream, g, z, M, N, stats, gamma,
                       part, chunks, pstride, dgamma, dbeta, extra_out);
    return re_launch_status();
}

extern "C" int re_bn_relu_drop_bwd(const float* da, const float* a, const float* z, int64_t M, int64_t N, const float* gamma,
                                   const float* stats, float drop_p, float* dz, float* dgamma, float* dbeta, void* ws, size_t ws_bytes,
                                   re_stream_t stream) {
    re_clear_error();
    if (!da || !a || !z || !dz || !dbeta || M <= 0 || N <= 0) return RE_EINVAL;
    if (!ws || ws_bytes < re_mlp_workspace_bytes(N)) return RE_EWORKSPACE;
    const bool bn = gamma != nullptr;
    if (bn && (!stats || !dgamma)) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int ch = ml_chunks(M);
    hipLaunchKernelGGL(bn_bwd_reduce_k, dim3((unsigned)re_cdiv(N, 64), ch), dim3(256), 0, s, da, a, z, M, N, bn ? stats : nullptr, ds, dz, (float*)ws);
    hipLaunchKernelGGL(col_final_k, dim3((unsigned)re_cdiv(N, 256)), dim3(256), 0, s, (const float*)ws, ch, N, dbeta, bn ? dgamma : nullptr);
    if (bn) hipLaunchKernelGGL(bn_bwd_apply_k, dim3(re_grid(M * N, 1024)), dim3(256), 0, s, dz, z, M * N, M, N, stats, gamma, (const float*)dgamma, (const float*)dbeta);
    return re_launch_status();
}

extern "C" int re_colsum(const float* x, int64_t M, int64_t N, float* out, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!x || !out || M <= 0 || N <= 0) return RE_EINVAL;
    if (!ws || ws_bytes < re_mlp_workspace_bytes(N)) return RE_EWORKSPACE;
    const int ch = ml_chunks(M);
    hipLaunchKernelGGL(colsum_k, dim3((unsigned)re_cdiv(N, 64), ch), dim3(256), 0, (hipStream_t)stream, x, M, N, (float*)ws);
    hipLaunchKernelGGL(col_final_k, dim3((unsigned)re_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)ws, ch, N, out, (float*)nullptr);
    return re_launch_status();
}


// ---------------------------------------------------------------------------------------------------------------------------------------
// DeepFM's last layer and criterion as two small launches each way (DeepFM/main.py:151-164 `dnn` ends in Linear(., 1); :201-215 logits =
// lr + fm + dnn, BCELoss4Logits mean).  As GEMMs these were N = 1 / K = 1 products on 64 x 64 tiles (18 us each for 6.5 MB of traffic) plus
// an elementwise add, the criterion and two copies.
//   forward:  logit[m] = fm_lr[m] + <h[m, :], w> + b   (natural-k fmaf chain; a lane group of 16 per row, four rows in flight)
//             labels given: dlogit[m] = (sigmoid(logit) - y) / M, loss = mean BCE, dsum = sum dlogit  (per-workgroup partials in a
//             fixed order, summed in order by a one-workgroup second launch)
//   backward: da[m, k] = dlogit[m] w[k];  dW[k] = sum_m dlogit[m] h[m, k]  (column sums by row chunk, then col_final_k)
#define HD_ROWS 16      // rows per workgroup of the forward (256 workgroups at M = 4096)
__global__ __launch_bounds__(256) void mlp_head_fwd_k(const float* __restrict__ h, int64_t M, int64_t K, const float* __restrict__ w,
                                                      const float* __restrict__ b, const float* __restrict__ fm_lr, const float* __restrict__ labels,
                                                      float* __restrict__ logits, float* __restrict__ dlogit, float* __restrict__ partial) {
    __shared__ float s_l[4], s_g[4];
    const int lane16 = threadIdx.x & 15, grp = threadIdx.x >> 4;          // 16 lane groups of 16 lanes: a row each, four trips
    const float inv = 1.0f / (float)M;
    float lsum = 0.f, gsum = 0.f;
#pragma unroll
    for (int t = 0; t < HD_ROWS / 16; ++t) {
        const int64_t m = (int64_t)blockIdx.x * HD_ROWS + t * 16 + grp;
        float acc = 0.f;
        if (m < M)
            for (int64_t k = lane16 * 4; k < K; k += 64) {                // (K a multiple of 4: the launcher checks)
                const float4 a = *reinterpret_cast<const float4*>(h + m * K + k), c = *reinterpret_cast<const float4*>(w + k);
                acc = fmaf(a.x, c.x, acc); acc = fmaf(a.y, c.y, acc); acc = fmaf(a.z, c.z, acc); acc = fmaf(a.w, c.w, acc);
            }
        // the row's sixteen partial dots, in a fixed tree
        acc += __shfl_xor(acc, 8, 16); acc += __shfl_xor(acc, 4, 16); acc += __shfl_xor(acc, 2, 16); acc += __shfl_xor(acc, 1, 16);
        if (m < M && lane16 == 0) {
            const float x = acc + b[0] + (fm_lr ? fm_lr[m] : 0.f);
            logits[m] = x;
            if (labels) {
                const float y = labels[m];
                lsum += fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
                const float d = (re_sigmoid(x) - y) * inv;
                dlogit[m] = d;
                gsum += d;
            }
        }
    }
    if (!labels) return;
    lsum = re_wave_sum(lsum); gsum = re_wave_sum(gsum);
    if ((threadIdx.x & 63) == 0) { s_l[threadIdx.x >> 6] = lsum; s_g[threadIdx.x >> 6] = gsum; }
    __syncthreads();
    if (threadIdx.x == 0) {
        partial[2 * blockIdx.x] = ((s_l[0] + s_l[1]) + s_l[2]) + s_l[3];
        partial[2 * blockIdx.x + 1] = ((s_g[0] + s_g[1]) + s_g[2]) + s_g[3];
    }
}
__global__ __launch_bounds__(64) void mlp_head_final_k(const float* __restrict__ partial, int nb, float inv, float* __restrict__ loss,
                                                       float* __restrict__ dsum, float* __restrict__ dsum2) {
    float a = 0.f, g = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) { a += partial[2 * i]; g += partial[2 * i + 1]; }
    a = re_wave_sum(a); g = re_wave_sum(g);
    if (threadIdx.x == 0) { loss[0] = a * inv; if (dsum) dsum[0] = g; if (dsum2) dsum2[0] = g; }
}
__global__ __launch_bounds__(256) void mlp_head_bwd_k(const float* __restrict__ dlogit, const float* __restrict__ h, const float* __restrict__ w,
                                                      int64_t M, int64_t K, float* __restrict__ da, float* __restrict__ partial) {
    float s, d;
    bool owner;
    int64_t col, mb, me;
    const int64_t myc = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const float wk = myc < K ? w[myc] : 0.f;
    col_reduce2(M, K, [&](int64_t m, int64_t k, float& a, float& b2) { a = dlogit[m] * h[m * K + k]; b2 = 0.f; }, s, d, owner, col, mb, me,
                [&](int64_t m, int64_t k, float) { da[m * K + k] = dlogit[m] * wk; });
    if (owner) {
        partial[((int64_t)blockIdx.y * 2 + 0) * K + col] = s;
        partial[((int64_t)blockIdx.y * 2 + 1) * K + col] = 0.f;
    }
}

extern "C" size_t re_mlp_head_workspace_bytes(int64_t M, int64_t K) {
    const size_t a = (size_t)re_cdiv(M, HD_ROWS) * 2 * sizeof(float), b = (size_t)ML_CHUNKS * 3 * K * sizeof(float);
    return (a > b ? a : b) + 256;
}
// labels == NULL: the logits alone (evaluation); otherwise loss [1], dlogit [M], and sum dlogit into dsum [1] and dsum2 [1] (either may be
// null: DeepFM has two biases with that gradient, the last layer's and the LR term's) too.
extern "C" int re_mlp_head_fwd(const float* h, int64_t M, int64_t K, const float* w, const float* b, const float* fm_lr, const float* labels,
                               float* logits, float* loss, float* dlogit, float* dsum, float* dsum2, void* ws, size_t ws_bytes,
                               re_stream_t stream) {
    re_clear_error();
    if (!h || !w || !b || !logits || M <= 0 || K <= 0) return RE_EINVAL;
    if (labels && !dlogit) return RE_EINVAL;      // (loss == NULL with labels: the per-workgroup partials stay in ws -- re_mlp_head_bwd_gated sums them)
    if ((K & 3) || ((reinterpret_cast<uintptr_t>(h) | reinterpret_cast<uintptr_t>(w)) & 15u)) return RE_EUNSUPPORTED;
    if (labels && (!ws || ws_bytes < re_mlp_head_workspace_bytes(M, K))) return RE_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)re_cdiv(M, HD_ROWS);
    hipLaunchKernelGGL(mlp_head_fwd_k, dim3(nb), dim3(256), 0, s, h, M, K, w, b, fm_lr, labels, logits, dlogit, (float*)ws);
    if (labels && loss) hipLaunchKernelGGL(mlp_head_final_k, dim3(1), dim3(64), 0, s, (const float*)ws, nb, 1.0f / (float)M, loss, dsum, dsum2);
    return re_launch_status();
}
extern "C" int re_mlp_head_bwd(const float* dlogit, const float* h, const float* w, int64_t M, int64_t K, float* da, float* dW, void* ws,
                               size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!dlogit || !h || !w || !da || !dW || M <= 0 || K <= 0) return RE_EINVAL;
    if (!ws || ws_bytes < re_mlp_head_workspace_bytes(M, K)) return RE_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const int ch = ml_chunks(M);
    hipLaunchKernelGGL(mlp_head_bwd_k, dim3((unsigned)re_cdiv(K, 64), ch), dim3(256), 0, s, dlogit, h, w, M, K, da, (float*)ws);
    hipLaunchKernelGGL(col_final_k, dim3((unsigned)re_cdiv(K, 256)), dim3(256), 0, s, (const float*)ws, ch, K, dW, (float*)nullptr);
    return re_launch_status();
}

// re_mlp_head_bwd with the gate of the block underneath in the same pass (h is that block's output: positive exactly where relu passed and
// dropout kept): g [M, K] = h > 0 ? drop_scale dlogit[m] w[k] : 0, and per-row-chunk partials part [chunks][3][K] = (sum g, sum g xhat,
// sum dlogit h) -- re_bn_bwd_apply (pstride 3, extra_out = dW) finishes both.
__global__ __launch_bounds__(256) void mlp_head_bwd_gated_k(const float* __restrict__ dlogit, const float* __restrict__ h, const float* __restrict__ w,
                                                            const float* __restrict__ z, const float* __restrict__ stats, float drop_scale,
                                                            int64_t M, int64_t K, float* __restrict__ g, float* __restrict__ part,
                                                            const float* __restrict__ fin_partial, int fin_nb, float fin_inv,
                                                            float* __restrict__ loss, float* __restrict__ dsum, float* __restrict__ dsum2) {
    __shared__ float r0[256], r1[256], r2[256];
    // (the criterion's second launch -- mlp_head_final_k: the forward's per-workgroup (loss, sum dlogit) partials added in order -- rides here,
    //  in the first wave of the first workgroup: the same arithmetic, one dispatch less between the forward's head and its backward)
    if (fin_partial && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        float a = 0.f, gs = 0.f;
        for (int i = threadIdx.x; i < fin_nb; i += 64) { a += fin_partial[2 * i]; gs += fin_partial[2 * i + 1]; }
        a = re_wave_sum(a); gs = re_wave_sum(gs);
        if (threadIdx.x == 0) { loss[0] = a * fin_inv; if (dsum) dsum[0] = gs; if (dsum2) dsum2[0] = gs; }
    }
    const int c = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int64_t rpc = (M + gridDim.y - 1) / gridDim.y;
    const int64_t m_begin = (int64_t)blockIdx.y * rpc, m_end = (m_begin + rpc < M) ? m_begin + rpc : M;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    if (col < K) {
        const float wk = w[col] * drop_scale, mu = stats[col], rs = stats[K + col];
        auto one = [&](int64_t m, float dl, float hv, float zv) {
            const float gv = hv > 0.f ? dl * wk : 0.f;
            g[m * K + col] = gv;
            s0 += gv;
            s1 = fmaf(gv, (zv - mu) * rs, s1);
            s2 = fmaf(dl, hv, s2);
        };
        int64_t m = m_begin + rg;
        for (; m + 28 < m_end; m += 32) {
            float dl[8], hv[8], zv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { dl[i] = dlogit[m + 4 * i]; hv[i] = h[(m + 4 * i) * K + col]; zv[i] = z[(m + 4 * i) * K + col]; }
#pragma unroll
            for (int i = 0; i < 8; ++i) one(m + 4 * i, dl[i], hv[i], zv[i]);
        }
        for (; m < m_end; m += 4) one(m, dlogit[m], h[m * K + col], z[m * K + col]);
    }
    r0[threadIdx.x] = s0; r1[threadIdx.x] = s1; r2[threadIdx.x] = s2;
    __syncthreads();
    if (rg == 0 && col < K) {
        part[((int64_t)blockIdx.y * 3 + 0) * K + col] = ((r0[c] + r0[64 + c]) + r0[128 + c]) + r0[192 + c];
        part[((int64_t)blockIdx.y * 3 + 1) * K + col] = ((r1[c] + r1[64 + c]) + r1[128 + c]) + r1[192 + c];
        part[((int64_t)blockIdx.y * 3 + 2) * K + col] = ((r2[c] + r2[64 + c]) + r2[128 + c]) + r2[192 + c];
    }
}
// -> *chunks_out = the number of row chunks in `part` (part: >= re_mlp_head_workspace_bytes(M, K) bytes)
extern "C" int re_mlp_head_bwd_gated(const float* dlogit, const float* h, const float* w, int64_t M, int64_t K, const float* z,
                                     const float* stats, float drop_p, float* g, float* part, size_t part_bytes, int* chunks_out,
                                     const void* head_ws, float* loss, float* dsum, float* dsum2, re_stream_t stream) {
    re_clear_error();
    if (!dlogit || !h || !w || !z || !stats || !g || !part || !chunks_out || M <= 0 || K <= 0 || drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    if (head_ws && !loss) return RE_EINVAL;     // (head_ws: the workspace a re_mlp_head_fwd call with labels and loss == NULL left its partials in)
    if (part_bytes < re_mlp_head_workspace_bytes(M, K)) return RE_EWORKSPACE;
    const int ch = ml_chunks(M);
    hipLaunchKernelGGL(mlp_head_bwd_gated_k, dim3((unsigned)re_cdiv(K, 64), ch), dim3(256), 0, (hipStream_t)stream, dlogit, h, w, z, stats,
                       drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f, M, K, g, part, (const float*)head_ws, (int)re_cdiv(M, HD_ROWS), 1.0f / (float)M, loss,
                       dsum, dsum2);
    *chunks_out = ch;
    return re_launch_status();
}
