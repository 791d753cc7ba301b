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

template <int LPR, int ILP = RE_GATHER_ILP, bool NT = false, bool NTL = false>  // lanes per row; D = 4 * LPR * k
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
                if (r[u] >= 0 && r[u] < R) {
                    const float4* src = reinterpret_cast<const float4*>(W + r[u] * D) + c;
                    if (NTL) v[u] = make_float4(__builtin_nontemporal_load(&src->x), __builtin_nontemporal_load(&src->y),
                                                __builtin_nontemporal_load(&src->z), __builtin_nontemporal_load(&src->w));
                    else v[u] = *src;
                }
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
        // measured with scripts/tune_gather.py) and their rows read with non-temporal loads (another +2 %: 5.77 -> 5.88 TB/s);
        // small outputs stay cacheable for the consumer kernel.
        const bool nt = (size_t)n * D * sizeof(float) > ((size_t)32 << 20);
        const int64_t cap = 65536;
#define RE_GATHER_LAUNCH(LPRV)                                                                                              \
    do {                                                                                                                    \
        if (nt) hipLaunchKernelGGL((gather_rows_vec4<LPRV, RE_GATHER_ILP, true, true>), dim3(re_grid(n, (256 / LPRV) * RE_GATHER_ILP, cap)), dim3(256), 0, s, W, R, D, idx, n, out); \
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
// One workgroup per 8 consecutive float4 columns (128 B) of the [B][S*D] gradient: thread (tb, cq) walks rows b = tb, tb+32, ...
// of float4 column cq (4 rows in flight), applies the pad mask / dropout mask / sqrt(D) scale in place and keeps the
// unscaled column sum; the 32 row-lane partials are then added in lane order through LDS.  No workspace, no second kernel,
// and a fixed summation order (deterministic).
__global__ __launch_bounds__(256) void sasrec_embed_bwd_k(float* __restrict__ gx, const int64_t* __restrict__ seq, int B, int S,
                                                          int D, float scale, float drop_scale, uint32_t thresh, uint32_t seed,
                                                          float* __restrict__ dP, const uint32_t* __restrict__ seed_dev) {
    if (seed_dev) seed ^= seed_dev[0];
    __shared__ float4 red[32][8];
    const int cq = threadIdx.x & 7, tb = threadIdx.x >> 3;
    const int nf4 = S * D / 4, d4n = D / 4;
    const int f = blockIdx.x * 8 + cq;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < nf4) {
        const int s = f / d4n;
        float4* col = reinterpret_cast<float4*>(gx) + f;
        for (int b0 = tb; b0 < B; b0 += 128) {
            float4 v[4];
            int64_t sq[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + 32 * u;
                const bool ok = b < B;
                v[u] = ok ? col[(int64_t)b * nf4] : make_float4(0.f, 0.f, 0.f, 0.f);
                sq[u] = ok ? seq[(int64_t)b * S + s] : 0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + 32 * u;
                if (b >= B) break;
                float4 w = v[u];
                if (sq[u] == 0) w = make_float4(0.f, 0.f, 0.f, 0.f);
                if (thresh) {
                    const uint32_t e = (uint32_t)(((int64_t)b * nf4 + f) * 4);
                    w.x = re_keep(seed, RE_STREAM_EMBED, e + 0, thresh) ? w.x * drop_scale : 0.f;
                    w.y = re_keep(seed, RE_STREAM_EMBED, e + 1, thresh) ? w.y * drop_scale : 0.f;
                    w.z = re_keep(seed, RE_STREAM_EMBED, e + 2, thresh) ? w.z * drop_scale : 0.f;
                    w.w = re_keep(seed, RE_STREAM_EMBED, e + 3, thresh) ? w.w * drop_scale : 0.f;
                }
                acc.x += w.x; acc.y += w.y; acc.z += w.z; acc.w += w.w;
                col[(int64_t)b * nf4] = make_float4(w.x * scale, w.y * scale, w.z * scale, w.w * scale);
            }
        }
    }
    red[tb][cq] = acc;
    __syncthreads();
    if (tb == 0 && f < nf4) {
        float4 t = red[0][cq];
        for (int i = 1; i < 32; ++i) {
            const float4 r = red[i][cq];
            t.x += r.x; t.y += r.y; t.z += r.z; t.w += r.w;
        }
        reinterpret_cast<float4*>(dP)[f] = t;
    }
}

extern "C" size_t re_sasrec_embed_bwd_workspace_bytes(int64_t S, int64_t D) {
    (void)S; (void)D;
    return 256;   // (no workspace needed any more; the entry point keeps its ws arguments)
}

extern "C" int re_sasrec_embed_bwd(float* gx, const int64_t* seq, int64_t B, int64_t S, int64_t D, float scale, float drop_p,
                                   uint32_t seed, const uint32_t* seed_dev, float* dP, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    (void)ws; (void)ws_bytes;
    if (!gx || !seq || !dP || B < 0 || S <= 0 || D <= 0) return RE_EINVAL;
    if ((D & 3) || !aligned16(gx) || !aligned16(dP) || B * S * D >= 0x100000000ll) return RE_EUNSUPPORTED;
    if (drop_p < 0.f || drop_p >= 1.f) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t thresh = drop_p > 0.f ? re_drop_threshold(drop_p) : 0u;
    const float ds = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int nf4 = (int)(S * D / 4);
    hipLaunchKernelGGL(sasrec_embed_bwd_k, dim3((nf4 + 7) / 8), dim3(256), 0, s, gx, seq, (int)B, (int)S, (int)D, scale, ds, thresh, seed, dP,
                       seed_dev);
    return re_launch_status();
}

#ifdef RE_DEBUG
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
        case 6: hipLaunchKernelGGL((gather_rows_vec4<16, 4, true, true>), dim3(re_grid(n, 16 * 4, gridcap)), dim3(256), 0, s, W, R, D, idx, n, out); break;
        case 7: hipLaunchKernelGGL((gather_rows_vec4<16, 8, true, true>), dim3(re_grid(n, 16 * 8, gridcap)), dim3(256), 0, s, W, R, D, idx, n, out); break;
        default: return RE_EINVAL;
    }
    return re_launch_status();
}
#endif
