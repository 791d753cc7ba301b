// K1: embedding row gather (+ the fused SASRec front end).  HBM-bound.
//
// Layout: tables are row-major fp32 [R, D].  A row of D floats is read by D/4 lanes as float4 (D=64: 16 lanes
// x 16 B = one 256-B row, 4 rows per wave-instruction; D=128: 32 lanes).  Each lane group keeps RE_GATHER_ILP rows
// in flight (independent index loads, then independent row loads, then stores) so that a wave has
// 4 KB outstanding -- random-row gathers are latency-bound unless many loads are in flight
// (MI355X_MICROARCH.md "Indexed rows").  Rows that are not a multiple of 4 floats (DeepFM: D=10, D=1) take the
// element-per-lane path.
//
// Algorithmic bytes per looked-up row: 8 (index) + 4D (read) + 4D (write)  (SURVEY.md §8d).
#include "re_common.h"
#include "re_rng.h"

#define RE_GATHER_ILP 4

template <int LPR>  // lanes per row; D = 4 * LPR * k
__global__ __launch_bounds__(256) void gather_rows_vec4(const float* __restrict__ W, int64_t R, int64_t D,
                                                        const int64_t* __restrict__ idx, int64_t n,
                                                        float* __restrict__ out) {
    const int lane_in_row = threadIdx.x % LPR;
    const int64_t groups_per_block = 256 / LPR;
    const int64_t group = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / LPR;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    const int64_t D4 = D >> 2;
    for (int64_t base = group * RE_GATHER_ILP; base < n; base += ngroups * RE_GATHER_ILP) {
        int64_t r[RE_GATHER_ILP];
#pragma unroll
        for (int u = 0; u < RE_GATHER_ILP; ++u) {
            int64_t i = base + u;
            r[u] = i < n ? idx[i] : -1;
        }
        for (int64_t c = lane_in_row; c < D4; c += LPR) {
            float4 v[RE_GATHER_ILP];
#pragma unroll
            for (int u = 0; u < RE_GATHER_ILP; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r[u] >= 0 && r[u] < R) v[u] = reinterpret_cast<const float4*>(W + r[u] * D)[c];
            }
#pragma unroll
            for (int u = 0; u < RE_GATHER_ILP; ++u) {
                int64_t i = base + u;
                if (i < n) reinterpret_cast<float4*>(out + i * D)[c] = v[u];
            }
        }
    }
}

__global__ __launch_bounds__(256) void gather_rows_scalar(const float* __restrict__ W, int64_t R, int64_t D,
                                                          const int64_t* __restrict__ idx, int64_t n,
                                                          float* __restrict__ out) {
    const int64_t total = n * D;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        int64_t i = e / D, d = e - i * D;
        int64_t r = idx[i];
        out[e] = (r >= 0 && r < R) ? W[r * D + d] : 0.0f;
    }
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" int re_gather_rows(const float* W, int64_t R, int64_t D, const int64_t* idx, int64_t n, float* out,
                              re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (!W || !idx || !out || R <= 0 || D <= 0 || n < 0) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if ((D & 3) == 0 && aligned16(W) && aligned16(out)) {
        const int64_t D4 = D >> 2;
        if (D4 >= 32) {
            hipLaunchKernelGGL(gather_rows_vec4<32>, dim3(re_grid(n, (256 / 32) * RE_GATHER_ILP)), dim3(256), 0, s, W, R, D, idx, n, out);
        } else if (D4 >= 16) {
            hipLaunchKernelGGL(gather_rows_vec4<16>, dim3(re_grid(n, (256 / 16) * RE_GATHER_ILP)), dim3(256), 0, s, W, R, D, idx, n, out);
        } else if (D4 >= 8) {
            hipLaunchKernelGGL(gather_rows_vec4<8>, dim3(re_grid(n, (256 / 8) * RE_GATHER_ILP)), dim3(256), 0, s, W, R, D, idx, n, out);
        } else {
            hipLaunchKernelGGL(gather_rows_vec4<4>, dim3(re_grid(n, (256 / 4) * RE_GATHER_ILP)), dim3(256), 0, s, W, R, D, idx, n, out);
        }
    } else {
        hipLaunchKernelGGL(gather_rows_scalar, dim3(re_grid(n * D, 256 * 4)), dim3(256), 0, s, W, R, D, idx, n, out);
    }
    return re_launch_status();
}

// out[b,s,:] = seq==0 ? 0 : dropout(E[seq]*scale + P[s])      (SASRec/main.py:181-187)
template <int LPR>
__global__ __launch_bounds__(256) void sasrec_embed_vec4(const float* __restrict__ E, int64_t R, int64_t D,
                                                         const float* __restrict__ P,
                                                         const int64_t* __restrict__ seq, int64_t n, int64_t S,
                                                         float scale, float drop_scale, uint32_t thresh,
                                                         uint32_t seed, float* __restrict__ out) {
    const int lane_in_row = threadIdx.x % LPR;
    const int64_t groups_per_block = 256 / LPR;
    const int64_t group = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / LPR;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    const int64_t D4 = D >> 2;
    for (int64_t base = group * RE_GATHER_ILP; base < n; base += ngroups * RE_GATHER_ILP) {
        int64_t r[RE_GATHER_ILP];
#pragma unroll
        for (int u = 0; u < RE_GATHER_ILP; ++u) {
            int64_t i = base + u;
            r[u] = i < n ? seq[i] : 0;
        }
        for (int64_t c = lane_in_row; c < D4; c += LPR) {
            float4 v[RE_GATHER_ILP], p[RE_GATHER_ILP];
#pragma unroll
            for (int u = 0; u < RE_GATHER_ILP; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                p[u] = v[u];
                int64_t i = base + u;
                if (r[u] > 0 && r[u] < R) {
                    v[u] = reinterpret_cast<const float4*>(E + r[u] * D)[c];
                    p[u] = reinterpret_cast<const float4*>(P + (i % S) * D)[c];
                }
            }
#pragma unroll
            for (int u = 0; u < RE_GATHER_ILP; ++u) {
                int64_t i = base + u;
                if (i >= n) continue;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r[u] > 0 && r[u] < R) {
                    o.x = v[u].x * scale + p[u].x;
                    o.y = v[u].y * scale + p[u].y;
                    o.z = v[u].z * scale + p[u].z;
                    o.w = v[u].w * scale + p[u].w;
                    if (thresh) {
                        uint32_t e = (uint32_t)(i * D + c * 4);
                        o.x = re_keep(seed, RE_STREAM_EMBED, e + 0, thresh) ? o.x * drop_scale : 0.f;
                        o.y = re_keep(seed, RE_STREAM_EMBED, e + 1, thresh) ? o.y * drop_scale : 0.f;
                        o.z = re_keep(seed, RE_STREAM_EMBED, e + 2, thresh) ? o.z * drop_scale : 0.f;
                        o.w = re_keep(seed, RE_STREAM_EMBED, e + 3, thresh) ? o.w * drop_scale : 0.f;
                    }
                }
                reinterpret_cast<float4*>(out + i * D)[c] = o;
            }
        }
    }
}

extern "C" int re_sasrec_embed(const float* E, int64_t R, int64_t D, const float* P, const int64_t* seq, int64_t B,
                               int64_t S, float scale, float drop_p, uint32_t seed, float* out, re_stream_t stream) {
    re_clear_error();
    const int64_t n = B * S;
    if (n == 0) return RE_OK;
    if (!E || !P || !seq || !out || R <= 0 || D <= 0 || B < 0 || S <= 0) return RE_EINVAL;
    if ((D & 3) != 0 || !aligned16(E) || !aligned16(P) || !aligned16(out)) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    if ((D >> 2) >= 32)
        hipLaunchKernelGGL(sasrec_embed_vec4<32>, dim3(re_grid(n, 8 * RE_GATHER_ILP)), dim3(256), 0, s, E, R, D, P, seq, n, S, scale, ds, thresh, seed, out);
    else
        hipLaunchKernelGGL(sasrec_embed_vec4<16>, dim3(re_grid(n, 16 * RE_GATHER_ILP)), dim3(256), 0, s, E, R, D, P, seq, n, S, scale, ds, thresh, seed, out);
    return re_launch_status();
}
