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

template <int LPR, int ILP = RE_GATHER_ILP, bool NT = false>  // lanes per row; D = 4 * LPR * k
__global__ __launch_bounds__(256) void gather_rows_vec4(const float* __restrict__ W, int64_t R, int64_t D,
                                                        const int64_t* __restrict__ idx, int64_t n,
                                                        float* __restrict__ out) {
    const int lane_in_row = threadIdx.x % LPR;
    const int64_t groups_per_block = 256 / LPR;
    const int64_t group = (int64_t)blockIdx.x * groups_per_block + threadIdx.x / LPR;
    const int64_t ngroups = (int64_t)gridDim.x * groups_per_block;
    const int64_t D4 = D >> 2;
    for (int64_t base = group * ILP; base < n; base += ngroups * ILP) {
        int64_t r[ILP];
#pragma unroll
        for (int u = 0; u < ILP; ++u) {
            int64_t i = base + u;
            r[u] = i < n ? idx[i] : -1;
        }
        for (int64_t c = lane_in_row; c < D4; c += LPR) {
            float4 v[ILP];
#pragma unroll
            for (int u = 0; u < ILP; ++u) {
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r[u] >= 0 && r[u] < R) v[u] = reinterpret_cast<const float4*>(W + r[u] * D)[c];
            }
#pragma unroll
            for (int u = 0; u < ILP; ++u) {
                int64_t i = base + u;
                if (i < n) {
                    float4* dst = reinterpret_cast<float4*>(out + i * D) + c;
                    if (NT) {
                        __builtin_nontemporal_store(v[u].x, &dst->x); __builtin_nontemporal_store(v[u].y, &dst->y);
                        __builtin_nontemporal_store(v[u].z, &dst->z); __builtin_nontemporal_store(v[u].w, &dst->w);
                    } else {
                        *dst = v[u];
                    }
                }
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
        // Outputs larger than the L2s are streamed with non-temporal stores (+9 % on a 4 GiB-table gather: 5.2 -> 5.7 TB/s,
        // measured with scripts/tune_gather.py); small outputs stay cacheable for the consumer kernel.
        const bool nt = (size_t)n * D * sizeof(float) > ((size_t)32 << 20);
        const int64_t cap = 65536;
#define RE_GATHER_LAUNCH(LPRV)                                                                                              \
    do {                                                                                                                    \
        if (nt) hipLaunchKernelGGL((gather_rows_vec4<LPRV, RE_GATHER_ILP, true>), dim3(re_grid(n, (256 / LPRV) * RE_GATHER_ILP, cap)), dim3(256), 0, s, W, R, D, idx, n, out); \
        else hipLaunchKernelGGL((gather_rows_vec4<LPRV, RE_GATHER_ILP, false>), dim3(re_grid(n, (256 / LPRV) * RE_GATHER_ILP, cap)), dim3(256), 0, s, W, R, D, idx, n, out); \
    } while (0)
        if (D4 >= 32) RE_GATHER_LAUNCH(32);
        else if (D4 >= 16) RE_GATHER_LAUNCH(16);
        else if (D4 >= 8) RE_GATHER_LAUNCH(8);
        else RE_GATHER_LAUNCH(4);
#undef RE_GATHER_LAUNCH
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
                                                         uint32_t seed, float* __restrict__ out,
                                                         const uint32_t* __restrict__ seed_dev) {
    if (seed_dev) seed ^= seed_dev[0];   // per-step seed kept in device memory (hipGraph replays)
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
                               int64_t S, float scale, float drop_p, uint32_t seed, const uint32_t* seed_dev, float* out,
                               re_stream_t stream) {
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
        hipLaunchKernelGGL(sasrec_embed_vec4<32>, dim3(re_grid(n, 8 * RE_GATHER_ILP)), dim3(256), 0, s, E, R, D, P, seq, n, S, scale, ds, thresh, seed, out, seed_dev);
    else
        hipLaunchKernelGGL(sasrec_embed_vec4<16>, dim3(re_grid(n, 16 * RE_GATHER_ILP)), dim3(256), 0, s, E, R, D, P, seq, n, S, scale, ds, thresh, seed, out, seed_dev);
    return re_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// backward of the SASRec front end.  In place on gx [B,S,D]: in = gradient w.r.t. x0, out = the contribution rows for
// re_scatter_add_rows (pad rows zero, dropout mask re-applied, times `scale`).  dP[s,:] = sum_b masked gradient.
// Deterministic two-stage reduction: EB_WGS workgroups own b = w, w+EB_WGS, ... and keep their [S,D] partial in
// registers; a second kernel adds the partials in workgroup order.
#define EB_WGS 64
#define EB_SLOTS 4  // float4 slots per thread: S*D/4 <= 1024

__global__ __launch_bounds__(256) void sasrec_embed_bwd_k(float* __restrict__ gx, const int64_t* __restrict__ seq, int B, int S,
                                                          int D, float scale, float drop_scale, uint32_t thresh, uint32_t seed,
                                                          float* __restrict__ part, const uint32_t* __restrict__ seed_dev) {
    if (seed_dev) seed ^= seed_dev[0];
    const int nf4 = S * D / 4, d4n = D / 4;
    float4 acc[EB_SLOTS];
#pragma unroll
    for (int k = 0; k < EB_SLOTS; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        float4* row = reinterpret_cast<float4*>(gx + (int64_t)b * S * D);
#pragma unroll
        for (int k = 0; k < EB_SLOTS; ++k) {
            const int f = k * 256 + threadIdx.x;
            if (f >= nf4) continue;
            const int s = f / d4n;
            float4 v = row[f];
            if (seq[(int64_t)b * S + s] == 0) v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (thresh) {
                const uint32_t e = (uint32_t)(((int64_t)b * S * D) + (int64_t)f * 4);
                v.x = re_keep(seed, RE_STREAM_EMBED, e + 0, thresh) ? v.x * drop_scale : 0.f;
                v.y = re_keep(seed, RE_STREAM_EMBED, e + 1, thresh) ? v.y * drop_scale : 0.f;
                v.z = re_keep(seed, RE_STREAM_EMBED, e + 2, thresh) ? v.z * drop_scale : 0.f;
                v.w = re_keep(seed, RE_STREAM_EMBED, e + 3, thresh) ? v.w * drop_scale : 0.f;
            }
            acc[k].x += v.x; acc[k].y += v.y; acc[k].z += v.z; acc[k].w += v.w;
            row[f] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
        }
    }
#pragma unroll
    for (int k = 0; k < EB_SLOTS; ++k) {
        const int f = k * 256 + threadIdx.x;
        if (f < nf4) reinterpret_cast<float4*>(part + (int64_t)blockIdx.x * nf4 * 4)[f] = acc[k];
    }
}

__global__ __launch_bounds__(256) void sasrec_embed_bwd_reduce(const float* __restrict__ part, int nwg, int n, float* __restrict__ dP) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float s = 0.f;
    for (int w = 0; w < nwg; ++w) s += part[(int64_t)w * n + e];
    dP[e] = s;
}

extern "C" size_t re_sasrec_embed_bwd_workspace_bytes(int64_t S, int64_t D) { return (size_t)EB_WGS * S * D * sizeof(float) + 256; }

extern "C" int re_sasrec_embed_bwd(float* gx, const int64_t* seq, int64_t B, int64_t S, int64_t D, float scale, float drop_p,
                                   uint32_t seed, const uint32_t* seed_dev, float* dP, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!gx || !seq || !dP || !ws || B < 0 || S <= 0 || D <= 0) return RE_EINVAL;
    if ((D & 3) || S * D / 4 > 256 * EB_SLOTS || !aligned16(gx) || !aligned16(dP)) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    if (ws_bytes < re_sasrec_embed_bwd_workspace_bytes(S, D)) return RE_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int nwg = (int)(B < EB_WGS ? (B > 0 ? B : 1) : EB_WGS);
    hipLaunchKernelGGL(sasrec_embed_bwd_k, dim3(nwg), dim3(256), 0, s, gx, seq, (int)B, (int)S, (int)D, scale, ds, thresh, seed, (float*)ws, seed_dev);
    const int n = (int)(S * D);
    hipLaunchKernelGGL(sasrec_embed_bwd_reduce, dim3((n + 255) / 256), dim3(256), 0, s, (const float*)ws, nwg, n, dP);
    return re_launch_status();
}

// ---- tuning hook (not part of the ABI): D = 64 gather with a chosen rows-in-flight / store policy / grid cap
extern "C" int re_dbg_gather64(const float* W, int64_t R, const int64_t* idx, int64_t n, float* out, int variant, int gridcap,
                               re_stream_t stream) {
    re_clear_error();
    hipStream_t s = (hipStream_t)stream;
    const int64_t D = 64;
#define RE_DBG_LAUNCH(ILPV, NTV)                                                                                         \
    hipLaunchKernelGGL((gather_rows_vec4<16, ILPV, NTV>), dim3(re_grid(n, 16 * ILPV, gridcap)), dim3(256), 0, s, W, R, D, idx, n, out)
    switch (variant) {
        case 0: RE_DBG_LAUNCH(4, false); break;
        case 1: RE_DBG_LAUNCH(8, false); break;
        case 2: RE_DBG_LAUNCH(4, true); break;
        case 3: RE_DBG_LAUNCH(8, true); break;
        case 4: RE_DBG_LAUNCH(2, false); break;
        case 5: RE_DBG_LAUNCH(16, false); break;
        default: return RE_EINVAL;
    }
    return re_launch_status();
}
