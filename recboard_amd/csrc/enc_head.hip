// Loss head of the fused SASRec step on the plan's COMPACT rows (csrc/enc_common.h): the pair criteria of pairloss.hip
// (SASRec/main.py:199-215) evaluated only where a row exists.
//
// The reference masks the [B, S] positions with `indices = positives != 0`, a boolean compaction that synchronises with the host
// (SASRec/main.py:199-204); re_pair_loss_fwd_bwd walks all B*S positions and writes three gradient rows for each, 88 % of them
// zero rows of padding positions on Beauty-shaped batches.  The batch plan already lists the rows that exist (the positions from
// a sequence's first real token on), so this kernel walks those: per compact row r (position gid = rowmap[r].x)
//     pl = <u[gid], E[e_off + pos[gid]]>,  nl = <u[gid], E[e_off + neg[gid]]>,  l = BCE pair / BPR
//     dU_rows[r]        = dpl * E[pos] + dnl * E[neg]                 (zero row where seq[gid] == 0)
//     g_rows[1][r]      = dpl * u,   g_rows[2][r] = dnl * u           (region 0 is written by re_sasrec_encoder_bwd: dx0_rows)
//     keys[0..2][r]     = seq[gid] | e_off + pos[gid] | e_off + neg[gid]   destination rows of the three contribution rows, 0 = none
// -- the operands of re_scatter_add_rows_small.  Loss: one partial per workgroup (fixed row -> thread map) added into a 64-bit
// fixed-point accumulator (order-independent, so deterministic); the last workgroup to arrive writes the mean (no second launch).
#include "enc_common.h"

#define EH_BLOCKS 512

template <int LPR>
__device__ __forceinline__ float eh_group_sum(float v) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float eh_dot4(const float4& a, const float4& b) { return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x))); }

template <int LPR>   // lanes per row: D / 4
__global__ __launch_bounds__(256) void loss_rows_k(const float* __restrict__ U, const float* __restrict__ E, int64_t R, int64_t e_off,
                                                   const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                                   const int64_t* __restrict__ neg, const void* __restrict__ planp, int B, int S, int kind,
                                                   const int32_t* __restrict__ count, float* __restrict__ dUr, float* __restrict__ G,
                                                   int32_t* __restrict__ keys, unsigned long long* __restrict__ acc, unsigned* __restrict__ done,
                                                   float* __restrict__ loss) {
    constexpr int D = 4 * LPR, GPB = 256 / LPR;
    __shared__ float s_sum[4];
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int nr = 16 * PL.hdr[1];
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    const int lir = threadIdx.x % LPR, grp = threadIdx.x / LPR;
    const float gs = 1.0f / (float)count[0];
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    float lsum = 0.f;
    for (int r0 = blockIdx.x * GPB; r0 < nr; r0 += gridDim.x * GPB) {
        const int r = r0 + grp;   // (nr is a multiple of 16 and so of GPB: every group has a row)
        const int gid = PL.rowmap[r].x;
        int64_t item = 0, pr = 0, ng = 0;
        if (gid >= 0) { item = seq[gid]; pr = pos[gid] + e_off; ng = neg[gid] + e_off; }
        const bool real = item > 0 && item < R;
        const bool ok = real && pr > 0 && pr < R && ng > 0 && ng < R;
        if (lir == 0) {
            keys[r] = real ? (int32_t)item : 0;
            keys[NR + r] = ok ? (int32_t)pr : 0;
            keys[2 * NR + r] = ok ? (int32_t)ng : 0;
        }
        float4* du = reinterpret_cast<float4*>(dUr + (int64_t)r * D);
        if (!ok) {   // (uniform over the lane group)
            du[lir] = z;
            continue;
        }
        const float4 a = reinterpret_cast<const float4*>(U + (int64_t)gid * D)[lir];
        const float4 b = reinterpret_cast<const float4*>(E + pr * D)[lir];
        const float4 d = reinterpret_cast<const float4*>(E + ng * D)[lir];
        const float pl = eh_group_sum<LPR>(eh_dot4(a, b)), nl = eh_group_sum<LPR>(eh_dot4(a, d));
        float dpl, dnl;
        if (kind == RE_LOSS_BCE) { dpl = -re_sigmoid(-pl) * gs; dnl = re_sigmoid(nl) * gs; }
        else { const float sg = re_sigmoid(nl - pl) * gs; dpl = -sg; dnl = sg; }
        du[lir] = make_float4(fmaf(dpl, b.x, dnl * d.x), fmaf(dpl, b.y, dnl * d.y), fmaf(dpl, b.z, dnl * d.z), fmaf(dpl, b.w, dnl * d.w));
        reinterpret_cast<float4*>(G + (NR + r) * D)[lir] = make_float4(dpl * a.x, dpl * a.y, dpl * a.z, dpl * a.w);
        reinterpret_cast<float4*>(G + (2 * NR + r) * D)[lir] = make_float4(dnl * a.x, dnl * a.y, dnl * a.z, dnl * a.w);
        if (lir == 0) lsum += (kind == RE_LOSS_BCE) ? re_softplus(-pl) + re_softplus(nl) : re_softplus(nl - pl);
    }
    lsum = re_wave_sum(lsum);
    const int wid = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_sum[wid] = lsum;
    __syncthreads();
    if (threadIdx.x == 0) {
        // The workgroup's partial goes into ONE 64-bit fixed-point accumulator (2^-30 units: integer adds commute, so the total does
        // not depend on the order the workgroups arrive in), then the workgroup takes a ticket; the last ticket holder reads the
        // total and writes the loss.  Both are device-scope atomics, ordered by a data dependency (the ticket's operand is made from
        // the add's return value) -- no fence: a release fence here writes back this XCD's whole L2, once per workgroup.
        const double part = (double)(((s_sum[0] + s_sum[1]) + s_sum[2]) + s_sum[3]);
        // (a NaN / Inf / out-of-range partial -- a diverged model -- is not converted: it is counted in the ticket word's upper bits
        //  (grids stay below 2^20 workgroups) and the finishing workgroup writes NaN, as torch's mean would)
        const bool finite = part == part && fabs(part) < 4294967296.0;
        const unsigned long long add = finite ? (unsigned long long)(long long)llrint(part * 1073741824.0) : 0ull;
        const unsigned long long old = __hip_atomic_fetch_add(acc, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned one = 1u + (finite ? 0u : (1u << 20)) + (unsigned)(old & 0ull);
        const unsigned ticket = __hip_atomic_fetch_add(done, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((ticket & 0xFFFFFu) == gridDim.x - 1) {
            const unsigned long long tot = __hip_atomic_exchange(acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool bad = ((ticket + one) >> 20) != 0u;
            const int c = count[0];
            loss[0] = (c > 0 && !bad) ? (float)((double)(long long)tot * (1.0 / 1073741824.0) / (double)c) : __builtin_nanf("");   // mean over an empty set is NaN, as torch's
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
    }
}

extern "C" int64_t re_sasrec_plan_rows(int64_t B, int64_t S) { return (B > 0 && S > 0) ? 16 * enc_plan_max_tiles(B, S) : 0; }
extern "C" size_t re_sasrec_loss_rows_workspace_bytes(void) { return 256; }

extern "C" int re_sasrec_loss_rows(const float* U, const float* E, int64_t R, int64_t D, int64_t e_off, const int64_t* seq,
                                   const int64_t* pos, const int64_t* neg, int64_t B, int64_t S, const void* plan, int kind,
                                   const int32_t* count, float* loss, float* dU_rows, float* g_rows, int32_t* keys, void* ws,
                                   size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!U || !E || !seq || !pos || !neg || !plan || !count || !loss || !dU_rows || !g_rows || !keys || !ws || B < 0 || S < 1 || R < 1)
        return RE_EINVAL;
    if (kind != RE_LOSS_BCE && kind != RE_LOSS_BPR) return RE_EINVAL;
    if (D != 64 && D != 128) return RE_EUNSUPPORTED;
    if (R >= 0x7FFFFFFFll) return RE_EUNSUPPORTED;
    if (ws_bytes < re_sasrec_loss_rows_workspace_bytes()) return RE_EWORKSPACE;
    if ((reinterpret_cast<uintptr_t>(U) | reinterpret_cast<uintptr_t>(E) | reinterpret_cast<uintptr_t>(dU_rows) | reinterpret_cast<uintptr_t>(g_rows)) & 15u)
        return RE_EUNSUPPORTED;
    unsigned long long* acc = (unsigned long long*)ws;         // zero before the first call; every call leaves both words zero
    unsigned* done = (unsigned*)((char*)ws + 8);
    const int64_t mt = enc_plan_max_tiles(B, S);
    hipStream_t s = (hipStream_t)stream;
    if (D == 64) {
        const int grid = (int)(mt < EH_BLOCKS ? mt : EH_BLOCKS);   // 16 rows per workgroup round
        hipLaunchKernelGGL(loss_rows_k<16>, dim3(grid), dim3(256), 0, s, U, E, R, e_off, seq, pos, neg, plan, (int)B, (int)S, kind, count, dU_rows,
                           g_rows, keys, acc, done, loss);
    } else {
        const int grid = (int)(2 * mt < EH_BLOCKS ? 2 * mt : EH_BLOCKS);   // 8 rows per workgroup round
        hipLaunchKernelGGL(loss_rows_k<32>, dim3(grid), dim3(256), 0, s, U, E, R, e_off, seq, pos, neg, plan, (int)B, (int)S, kind, count, dU_rows,
                           g_rows, keys, acc, done, loss);
    }
    return re_launch_status();
}
