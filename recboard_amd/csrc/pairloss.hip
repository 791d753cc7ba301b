// K3: fused pair-logit criteria (BCE pair / BPR) over gathered table rows.  HBM-bound.
//
// One lane group (D/4 lanes, one float4 each per 4*LPR columns) per position: the user row and the two item
// rows are read once, both dots are reduced inside the lane group with DPP shuffles, softplus is evaluated in
// registers, and only the two logits are written (the backward needs nothing else).  The reference does this
// with 2 index kernels + 2 mul-sum + 2-3 elementwise + mean, and -- for SASRec -- a boolean
// mask compaction that synchronises with the host (SASRec/main.py:199-204); here the mask is an input and M is
// counted on device.
//
// Loss reduction is deterministic: per-thread partial (fixed position->thread map) -> wave shuffle tree ->
// per-block LDS -> one partial per block -> a single-wave finalize kernel sums the block partials in order.
//
// Algorithmic bytes per position: fwd 3*(8+4D) read + 8 write; bwd 3*4D read + 3*4D write.
#include "re_common.h"

#define PL_MAX_BLOCKS 1024

template <int LPR>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)));
}

// U rows: Ubase + urow*ldu with urow = uidx ? uidx[i] : i
template <int LPR>
__global__ __launch_bounds__(256) void pair_loss_fwd_k(const float* __restrict__ Ubase, int64_t ldu, int64_t RU,
                                                       const int64_t* __restrict__ uidx,
                                                       const float* __restrict__ E, int64_t R, int64_t D, int64_t e_off,
                                                       const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                                       const uint8_t* __restrict__ valid, int64_t n, int kind,
                                                       float* __restrict__ logits, float* __restrict__ bsum,
                                                       int32_t* __restrict__ bcnt) {
    __shared__ float s_sum[4];
    __shared__ int s_cnt[4];
    const int lir = threadIdx.x % LPR;
    const int64_t gpb = 256 / LPR;
    const int64_t D4 = D >> 2;
    float lsum = 0.f;
    int lcnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR; i < n; i += (int64_t)gridDim.x * gpb) {
        bool ok = valid ? valid[i] != 0 : true;
        int64_t ur = uidx ? uidx[i] : i;
        int64_t pr = pos[i] + e_off, nr = neg[i] + e_off;
        ok = ok && ur >= 0 && ur < RU && pr >= 0 && pr < R && nr >= 0 && nr < R;
        float pl = 0.f, nl = 0.f;
        if (ok) {
            const float4* u = reinterpret_cast<const float4*>(Ubase + ur * ldu);
            const float4* ep = reinterpret_cast<const float4*>(E + pr * D);
            const float4* en = reinterpret_cast<const float4*>(E + nr * D);
            for (int64_t c = lir; c < D4; c += LPR) {
                const float4 a = u[c], b = ep[c], d = en[c];
                pl += dot4(a, b);
                nl += dot4(a, d);
            }
        }
        pl = group_sum<LPR>(pl);
        nl = group_sum<LPR>(nl);
        if (lir == 0) {
            logits[2 * i] = pl;
            logits[2 * i + 1] = nl;
            if (ok) {
                lsum += (kind == RE_LOSS_BCE) ? re_softplus(-pl) + re_softplus(nl) : re_softplus(nl - pl);
                lcnt += 1;
            }
        }
    }
    lsum = re_wave_sum(lsum);
    lcnt = (int)re_wave_sum((float)lcnt);  // <= 2^24 per wave: exact in fp32
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wid] = lsum; s_cnt[wid] = lcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        bcnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

__global__ __launch_bounds__(64) void pair_loss_finalize(const float* __restrict__ bsum, const int32_t* __restrict__ bcnt,
                                                         int nblocks, float* __restrict__ loss, int32_t* __restrict__ count) {
    // one wave: lane l sums blocks l, l+64, ... in order, then a fixed shuffle tree -> deterministic
    const int lane = threadIdx.x;
    float s = 0.f;
    int c = 0;
    for (int b = lane; b < nblocks; b += 64) { s += bsum[b]; c += bcnt[b]; }
    s = re_wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    if (lane == 0) {
        loss[0] = c > 0 ? s / (float)c : 0.f / 0.f;  // mean over an empty set is NaN, as torch's
        if (count) count[0] = c;
    }
}

template <int LPR>
__global__ __launch_bounds__(256) void pair_loss_bwd_k(const float* __restrict__ Ubase, int64_t ldu, int64_t RU,
                                                       const int64_t* __restrict__ uidx,
                                                       const float* __restrict__ E, int64_t R, int64_t D, int64_t e_off,
                                                       const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                                       const uint8_t* __restrict__ valid, int64_t n, int kind,
                                                       const float* __restrict__ logits, const int32_t* __restrict__ count,
                                                       int64_t count_host, const float* __restrict__ dloss,
                                                       float* __restrict__ dU, int64_t lddu,
                                                       float* __restrict__ gpos, float* __restrict__ gneg) {
    const int lir = threadIdx.x % LPR;
    const int64_t gpb = 256 / LPR;
    const int64_t D4 = D >> 2;
    const float M = count ? (float)count[0] : (float)count_host;
    const float gs = (dloss ? dloss[0] : 1.0f) / M;
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int64_t i = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR; i < n; i += (int64_t)gridDim.x * gpb) {
        bool ok = valid ? valid[i] != 0 : true;
        int64_t ur = uidx ? uidx[i] : i;
        int64_t pr = pos[i] + e_off, nr = neg[i] + e_off;
        ok = ok && ur >= 0 && ur < RU && pr >= 0 && pr < R && nr >= 0 && nr < R;
        float4* du = reinterpret_cast<float4*>(dU + i * lddu);
        float4* gp = reinterpret_cast<float4*>(gpos + i * D);
        float4* gn = reinterpret_cast<float4*>(gneg + i * D);
        if (!ok) {
            for (int64_t c = lir; c < D4; c += LPR) { du[c] = z; gp[c] = z; gn[c] = z; }
            continue;
        }
        const float pl = logits[2 * i], nl = logits[2 * i + 1];
        float dpl, dnl;
        if (kind == RE_LOSS_BCE) { dpl = -re_sigmoid(-pl) * gs; dnl = re_sigmoid(nl) * gs; }
        else { const float sg = re_sigmoid(nl - pl) * gs; dpl = -sg; dnl = sg; }
        const float4* u = reinterpret_cast<const float4*>(Ubase + ur * ldu);
        const float4* ep = reinterpret_cast<const float4*>(E + pr * D);
        const float4* en = reinterpret_cast<const float4*>(E + nr * D);
        for (int64_t c = lir; c < D4; c += LPR) {
            const float4 a = u[c], b = ep[c], d = en[c];
            du[c] = make_float4(fmaf(dpl, b.x, dnl * d.x), fmaf(dpl, b.y, dnl * d.y), fmaf(dpl, b.z, dnl * d.z), fmaf(dpl, b.w, dnl * d.w));
            gp[c] = make_float4(dpl * a.x, dpl * a.y, dpl * a.z, dpl * a.w);
            gn[c] = make_float4(dnl * a.x, dnl * a.y, dnl * a.z, dnl * a.w);
        }
    }
}

// Forward and backward in ONE pass over the rows (training steps, where the upstream gradient of the mean loss is 1 and
// the number of valid positions M is known before the launch: a host count, or a device word computed at batch assembly).
// Reads each of the three rows once, writes the three gradient rows; the loss partials go through the same
// deterministic per-block reduction as the forward-only kernel.
template <int LPR>
__global__ __launch_bounds__(256) void pair_loss_fused_k(const float* __restrict__ Ubase, int64_t ldu, int64_t RU,
                                                         const int64_t* __restrict__ uidx,
                                                         const float* __restrict__ E, int64_t R, int64_t D, int64_t e_off,
                                                         const int64_t* __restrict__ pos, const int64_t* __restrict__ neg,
                                                         const uint8_t* __restrict__ valid, int64_t n, int kind,
                                                         const int32_t* __restrict__ count, int64_t count_host,
                                                         float* __restrict__ dU, int64_t lddu, float* __restrict__ gpos,
                                                         float* __restrict__ gneg, float* __restrict__ bsum, int32_t* __restrict__ bcnt,
                                                         int32_t* __restrict__ keys, int32_t key_off) {
    // keys (optional, int32 [3][n]): the destination rows of the three gradient-row sets in ONE table that holds the user rows first and the
    // item rows behind them (key_off = number of user rows); -1 for a row that contributes nothing -- what re_scatter_adam_rows_small takes
    __shared__ float s_sum[4];
    __shared__ int s_cnt[4];
    const int lir = threadIdx.x % LPR;
    const int64_t gpb = 256 / LPR;
    const int64_t D4 = D >> 2;
    const float gs = 1.0f / (count ? (float)count[0] : (float)count_host);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float lsum = 0.f;
    int lcnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * gpb + threadIdx.x / LPR; i < n; i += (int64_t)gridDim.x * gpb) {
        bool ok = valid ? valid[i] != 0 : true;
        int64_t ur = uidx ? uidx[i] : i;
        int64_t pr = pos[i] + e_off, nr = neg[i] + e_off;
        ok = ok && ur >= 0 && ur < RU && pr >= 0 && pr < R && nr >= 0 && nr < R;
        float4* du = reinterpret_cast<float4*>(dU + i * lddu);
        float4* gp = reinterpret_cast<float4*>(gpos + i * D);
        float4* gn = reinterpret_cast<float4*>(gneg + i * D);
        if (keys && lir == 0) {
            keys[i] = ok ? (int32_t)ur : -1;
            keys[n + i] = ok ? key_off + (int32_t)pr : -1;
            keys[2 * n + i] = ok ? key_off + (int32_t)nr : -1;
        }
        if (!ok) {   // (uniform over the lane group)
            for (int64_t c = lir; c < D4; c += LPR) { du[c] = z; gp[c] = z; gn[c] = z; }
            continue;
        }
        const float4* u = reinterpret_cast<const float4*>(Ubase + ur * ldu);
        const float4* ep = reinterpret_cast<const float4*>(E + pr * D);
        const float4* en = reinterpret_cast<const float4*>(E + nr * D);
        if (D4 <= LPR) {   // the common case (D = 64: 16 lanes x float4): rows stay in registers between the two halves
            float4 a = z, b = z, d = z;
            if (lir < D4) { a = u[lir]; b = ep[lir]; d = en[lir]; }
            const float pl = group_sum<LPR>(dot4(a, b)), nl = group_sum<LPR>(dot4(a, d));
            float dpl, dnl;
            if (kind == RE_LOSS_BCE) { dpl = -re_sigmoid(-pl) * gs; dnl = re_sigmoid(nl) * gs; }
            else { const float sg = re_sigmoid(nl - pl) * gs; dpl = -sg; dnl = sg; }
            if (lir < D4) {
                du[lir] = make_float4(fmaf(dpl, b.x, dnl * d.x), fmaf(dpl, b.y, dnl * d.y), fmaf(dpl, b.z, dnl * d.z), fmaf(dpl, b.w, dnl * d.w));
                gp[lir] = make_float4(dpl * a.x, dpl * a.y, dpl * a.z, dpl * a.w);
                gn[lir] = make_float4(dnl * a.x, dnl * a.y, dnl * a.z, dnl * a.w);
            }
            if (lir == 0) {
                lsum += (kind == RE_LOSS_BCE) ? re_softplus(-pl) + re_softplus(nl) : re_softplus(nl - pl);
                lcnt += 1;
            }
        } else {
            float pl = 0.f, nl = 0.f;
            for (int64_t c = lir; c < D4; c += LPR) {
                const float4 a = u[c], b = ep[c], d = en[c];
                pl += dot4(a, b);
                nl += dot4(a, d);
            }
            pl = group_sum<LPR>(pl);
            nl = group_sum<LPR>(nl);
            float dpl, dnl;
            if (kind == RE_LOSS_BCE) { dpl = -re_sigmoid(-pl) * gs; dnl = re_sigmoid(nl) * gs; }
            else { const float sg = re_sigmoid(nl - pl) * gs; dpl = -sg; dnl = sg; }
            for (int64_t c = lir; c < D4; c += LPR) {
                const float4 a = u[c], b = ep[c], d = en[c];
                du[c] = make_float4(fmaf(dpl, b.x, dnl * d.x), fmaf(dpl, b.y, dnl * d.y), fmaf(dpl, b.z, dnl * d.z), fmaf(dpl, b.w, dnl * d.w));
                gp[c] = make_float4(dpl * a.x, dpl * a.y, dpl * a.z, dpl * a.w);
                gn[c] = make_float4(dnl * a.x, dnl * a.y, dnl * a.z, dnl * a.w);
            }
            if (lir == 0) {
                lsum += (kind == RE_LOSS_BCE) ? re_softplus(-pl) + re_softplus(nl) : re_softplus(nl - pl);
                lcnt += 1;
            }
        }
    }
    lsum = re_wave_sum(lsum);
    lcnt = (int)re_wave_sum((float)lcnt);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_sum[wid] = lsum; s_cnt[wid] = lcnt; }
    __syncthreads();
    if (threadIdx.x == 0) {
        bsum[blockIdx.x] = ((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3];
        bcnt[blockIdx.x] = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
}

static bool ok16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

extern "C" size_t re_pair_loss_workspace_bytes(int64_t n) {
    (void)n;
    return (size_t)PL_MAX_BLOCKS * 8 + 256;
}

static int pl_grid(int64_t n, int lpr) {
    int64_t g = re_cdiv(n, 256 / lpr);
    if (g < 1) g = 1;
    if (g > PL_MAX_BLOCKS) g = PL_MAX_BLOCKS;
    return (int)g;
}

static int pair_fwd(const float* Ubase, int64_t ldu, int64_t RU, const int64_t* uidx, const float* E, int64_t R, int64_t D,
                    int64_t e_off, const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                    float* logits, float* loss, int32_t* count, void* ws, size_t ws_bytes, hipStream_t s) {
    if (!Ubase || !E || !pos || !neg || !logits || !loss || !ws || n < 0 || D <= 0 || R <= 0) return RE_EINVAL;
    if ((D & 3) || (ldu & 3) || !ok16(Ubase) || !ok16(E)) return RE_EUNSUPPORTED;
    if (kind != RE_LOSS_BCE && kind != RE_LOSS_BPR) return RE_EINVAL;
    if (ws_bytes < re_pair_loss_workspace_bytes(n)) return RE_EWORKSPACE;
    float* bsum = (float*)ws;
    int32_t* bcnt = (int32_t*)((char*)ws + PL_MAX_BLOCKS * 4);
    int grid;
    if ((D >> 2) >= 32) {
        grid = pl_grid(n, 32);
        hipLaunchKernelGGL(pair_loss_fwd_k<32>, dim3(grid), dim3(256), 0, s, Ubase, ldu, RU, uidx, E, R, D, e_off, pos, neg, valid, n, kind, logits, bsum, bcnt);
    } else {
        grid = pl_grid(n, 16);
        hipLaunchKernelGGL(pair_loss_fwd_k<16>, dim3(grid), dim3(256), 0, s, Ubase, ldu, RU, uidx, E, R, D, e_off, pos, neg, valid, n, kind, logits, bsum, bcnt);
    }
    hipLaunchKernelGGL(pair_loss_finalize, dim3(1), dim3(64), 0, s, bsum, bcnt, grid, loss, count);
    return re_launch_status();
}

static int pair_bwd(const float* Ubase, int64_t ldu, int64_t RU, const int64_t* uidx, const float* E, int64_t R, int64_t D,
                    int64_t e_off, const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                    const float* logits, const int32_t* count, int64_t count_host, const float* dloss, float* dU, int64_t lddu,
                    float* gpos, float* gneg, hipStream_t s) {
    if (n == 0) return RE_OK;
    if (!Ubase || !E || !pos || !neg || !logits || !dU || !gpos || !gneg || n < 0 || D <= 0) return RE_EINVAL;
    if ((D & 3) || (ldu & 3) || (lddu & 3) || !ok16(Ubase) || !ok16(E) || !ok16(dU) || !ok16(gpos) || !ok16(gneg)) return RE_EUNSUPPORTED;
    if ((D >> 2) >= 32)
        hipLaunchKernelGGL(pair_loss_bwd_k<32>, dim3(re_grid(n, 8)), dim3(256), 0, s, Ubase, ldu, RU, uidx, E, R, D, e_off, pos, neg, valid, n, kind, logits, count, count_host, dloss, dU, lddu, gpos, gneg);
    else
        hipLaunchKernelGGL(pair_loss_bwd_k<16>, dim3(re_grid(n, 16)), dim3(256), 0, s, Ubase, ldu, RU, uidx, E, R, D, e_off, pos, neg, valid, n, kind, logits, count, count_host, dloss, dU, lddu, gpos, gneg);
    return re_launch_status();
}

extern "C" int re_pair_loss_fwd(const float* U, int64_t ldu, const float* E, int64_t R, int64_t D, int64_t e_off,
                                const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                                float* logits, float* loss, int32_t* count, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    return pair_fwd(U, ldu, n > 0 ? n : 1, nullptr, E, R, D, e_off, pos, neg, valid, n, kind, logits, loss, count, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int re_pair_loss_bwd(const float* U, int64_t ldu, const float* E, int64_t R, int64_t D, int64_t e_off,
                                const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                                const float* logits, const int32_t* count, const float* dloss, float* dU, int64_t lddu,
                                float* gpos, float* gneg, re_stream_t stream) {
    re_clear_error();
    if (!count) return RE_EINVAL;
    return pair_bwd(U, ldu, n > 0 ? n : 1, nullptr, E, R, D, e_off, pos, neg, valid, n, kind, logits, count, 0, dloss, dU, lddu, gpos, gneg, (hipStream_t)stream);
}

extern "C" int re_bpr_triplet_fwd(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D,
                                  const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                                  float* logits, float* loss, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!users) return RE_EINVAL;
    return pair_fwd(Ut, D, RU, users, It, RI, D, 0, pos, neg, nullptr, n, RE_LOSS_BPR, logits, loss, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int re_bpr_triplet_bwd(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D,
                                  const int64_t* users, const int64_t* pos, const int64_t* neg, int64_t n,
                                  const float* logits, const float* dloss, float* gu, float* gpos, float* gneg,
                                  re_stream_t stream) {
    re_clear_error();
    if (!users) return RE_EINVAL;
    return pair_bwd(Ut, D, RU, users, It, RI, D, 0, pos, neg, nullptr, n, RE_LOSS_BPR, logits, nullptr, n, dloss, gu, D, gpos, gneg, (hipStream_t)stream);
}

static int pair_fused(const float* Ubase, int64_t ldu, int64_t RU, const int64_t* uidx, const float* E, int64_t R, int64_t D,
                      int64_t e_off, const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                      const int32_t* count, int64_t count_host, float* loss, float* dU, int64_t lddu, float* gpos, float* gneg,
                      void* ws, size_t ws_bytes, hipStream_t s, int32_t* keys = nullptr, int32_t key_off = 0) {
    if (!Ubase || !E || !pos || !neg || !loss || !dU || !gpos || !gneg || !ws || n < 0 || D <= 0 || R <= 0) return RE_EINVAL;
    if ((D & 3) || (ldu & 3) || (lddu & 3) || !ok16(Ubase) || !ok16(E) || !ok16(dU) || !ok16(gpos) || !ok16(gneg)) return RE_EUNSUPPORTED;
    if (kind != RE_LOSS_BCE && kind != RE_LOSS_BPR) return RE_EINVAL;
    if (ws_bytes < re_pair_loss_workspace_bytes(n)) return RE_EWORKSPACE;
    float* bsum = (float*)ws;
    int32_t* bcnt = (int32_t*)((char*)ws + PL_MAX_BLOCKS * 4);
    int grid;
    if ((D >> 2) >= 32) {
        grid = pl_grid(n, 32);
        hipLaunchKernelGGL(pair_loss_fused_k<32>, dim3(grid), dim3(256), 0, s, Ubase, ldu, RU, uidx, E, R, D, e_off, pos, neg, valid, n, kind,
                           count, count_host, dU, lddu, gpos, gneg, bsum, bcnt, keys, key_off);
    } else {
        grid = pl_grid(n, 16);
        hipLaunchKernelGGL(pair_loss_fused_k<16>, dim3(grid), dim3(256), 0, s, Ubase, ldu, RU, uidx, E, R, D, e_off, pos, neg, valid, n, kind,
                           count, count_host, dU, lddu, gpos, gneg, bsum, bcnt, keys, key_off);
    }
    hipLaunchKernelGGL(pair_loss_finalize, dim3(1), dim3(64), 0, s, bsum, bcnt, grid, loss, (int32_t*)nullptr);
    return re_launch_status();
}

extern "C" int re_pair_loss_fwd_bwd(const float* U, int64_t ldu, const float* E, int64_t R, int64_t D, int64_t e_off,
                                    const int64_t* pos, const int64_t* neg, const uint8_t* valid, int64_t n, int kind,
                                    const int32_t* count, float* loss, float* dU, int64_t lddu, float* gpos, float* gneg, void* ws,
                                    size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!count) return RE_EINVAL;
    return pair_fused(U, ldu, n > 0 ? n : 1, nullptr, E, R, D, e_off, pos, neg, valid, n, kind, count, 0, loss, dU, lddu, gpos, gneg, ws,
                      ws_bytes, (hipStream_t)stream);
}

// The same with the gradient rows laid out for ONE owner-computes launch over the user | item arena: g [3][n][D] (user rows' / positives' /
// negatives' gradients) and keys int32 [3][n] = rows of a table that holds the RU user rows first and the RI item rows behind them
// (re_scatter_adam_rows_small: scatter-add + the dense Adam of every row in that one launch).  MF-BPR/main.py:81-93,116-123.
extern "C" int re_bpr_triplet_step_rows(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D, const int64_t* users,
                                        const int64_t* pos, const int64_t* neg, int64_t n, float* loss, float* g, int32_t* keys, void* ws,
                                        size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!users || !g || !keys) return RE_EINVAL;
    if (RU + RI >= 0x7FFFFFFFll) return RE_EUNSUPPORTED;
    return pair_fused(Ut, D, RU, users, It, RI, D, 0, pos, neg, nullptr, n, RE_LOSS_BPR, nullptr, n, loss, g, D, g + n * D, g + 2 * n * D, ws, ws_bytes,
                      (hipStream_t)stream, keys, (int32_t)RU);
}

extern "C" int re_bpr_triplet_fwd_bwd(const float* Ut, int64_t RU, const float* It, int64_t RI, int64_t D, const int64_t* users,
                                      const int64_t* pos, const int64_t* neg, int64_t n, float* loss, float* gu, float* gpos,
                                      float* gneg, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!users) return RE_EINVAL;
    return pair_fused(Ut, D, RU, users, It, RI, D, 0, pos, neg, nullptr, n, RE_LOSS_BPR, nullptr, n, loss, gu, D, gpos, gneg, ws, ws_bytes,
                      (hipStream_t)stream);
}
