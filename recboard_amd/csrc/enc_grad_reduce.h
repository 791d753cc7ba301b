// The reduction behind the encoder's weight-gradient jobs: the split partials of the L x 6 D x D matrix gradients in split order, the per-tile (or
// per-workgroup) slabs of the vector gradients in a fixed tree, the position table's chunk partials in chunk order -- and, optionally, the dense
// Adam of every element it finishes.  As a launch of its own (enc_grad_reduce_k, enc_wgrad.hip) or as the LAST jobs of the step tail's ticket
// queue (enc_tail.hip, round 5: one launch and its 7 us less per step).  A "virtual block" = 256 threads' worth of the old launch's grid:
//   [0, nmat_blocks): 256 elements of the matrix gradients each;  then nvec_blocks: 64 columns of one (block, vector);  then npos_blocks.
#pragma once
#include "enc_wgrad_job.h"

// optional: the dense Adam of every gradient element the reduction finishes (re_adam_fuse: arenas of one layout)
struct EncAdam {
    const float* gbase;
    float *p, *m, *v;
    const float* hyper;
    float b1, b2, omb1, omb2, eps, wd;
    const unsigned* gate;   // optional device word: non-zero = gradients are written, parameters and moments stay (a hand-over of this step timed out)
};
// What the end of an element's chain needs and does not depend on the partial sums -- the step scalars, the gate word, the element's parameter
// and moments -- is requested FIRST (eg_pre, unconditional loads), together with the partials: the launch is then one memory round trip deep
// instead of three (partials; scalars, on which a branch depended; parameter and moments).
struct EgPre {
    float ss, ib, p, m, v;
};
__device__ __forceinline__ EgPre eg_pre(const EncAdam& A, const float* d) {
    EgPre e{0.f, 0.f, 0.f, 0.f, 0.f};
    if (A.p) {
        const int64_t i = d - A.gbase;
        e.p = A.p[i]; e.m = A.m[i]; e.v = A.v[i];
        const unsigned gate_w = *(A.gate ? A.gate : reinterpret_cast<const unsigned*>(A.hyper));
        e.ss = A.hyper[0];
        e.ib = A.hyper[1];
        e.ib = (A.gate && gate_w != 0u) ? 0.f : e.ib;   // ({0, 0}: the caller gated this step off; gate: a hand-over of this step timed out)
    }
    return e;
}
__device__ __forceinline__ void eg_put(const EncAdam& A, float* d, float g, const EgPre& e) {
    *d = g;
    if (A.p && e.ib != 0.f) {
        const int64_t i = d - A.gbase;
        float pp = e.p, mm = e.m, vv = e.v;
        re_adam1(pp, mm, vv, g, A.b1, A.b2, A.omb1, A.omb2, e.ss, e.ib, A.eps, A.wd);     // (adam_vec4_dev's arithmetic)
        A.p[i] = pp; A.m[i] = mm; A.v[i] = vv;
    }
}

struct EncGradDst {
    float* p[SE_MAX_BLOCKS][14];  // per block: ABI order of the 12 block gradients, then g_last_w, g_last_b (last block only)
};

struct EgReduce {
    const float *part, *slab;
    int nwg;
    const void* planp;
    int B, S, D, L;
    EncGradDst dst;
    int nmat_blocks, nvec_blocks, npos_blocks;
    int nsplit;   // partials per matrix (enc_wgrad_job.h: wg_nsplit / wg_nsplit_tail)
    const float* ppart;
    float inv_scale;
    float* dPtab;
    int by_tile;
    EncAdam AD;
};

// position-table gradient: the position jobs' group partials (enc_wgrad_job.h: ppart[p][group][D]) in group order, / scale.
// vb in [0, npos_blocks), tid in [0, 256)
// COHERENT: the partials were written earlier in THIS launch by other workgroups (agent-scope stores, a done count): read them with agent-scope
// loads, past whatever this XCD's L2 holds
template <bool COHERENT = false>
__device__ __forceinline__ float eg_ld(const float* p) {
    if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool COHERENT = false>
__device__ __forceinline__ void eg_reduce_pos(const EgReduce& R, int vb, int tid, float (*red)[64]) {
    // 64 elements a block; wave q sums the partials of groups [36 q, 36 q + 36) (all requested together: one memory round trip), thread (0, c)
    // adds the four in order
    constexpr int PER = WG_POS_GROUPS / 4;
    static_assert(PER * 4 == WG_POS_GROUPS, "four waves share the groups");
    const int q = tid >> 6, e = vb * 64 + (tid & 63);
    const bool in = e < R.S * R.D;
    const int p = in ? e / R.D : 0, cc = in ? e % R.D : 0;
    float x[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) x[u] = eg_ld<COHERENT>(R.ppart + ((int64_t)p * WG_POS_GROUPS + q * PER + u) * R.D + cc);
    EgPre pre{0.f, 0.f, 0.f, 0.f, 0.f};
    if (q == 0) pre = eg_pre(R.AD, R.dPtab + (in ? e : 0));
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) s += x[u];
    red[q][tid & 63] = s;
    __syncthreads();
    if (q == 0 && in) eg_put(R.AD, R.dPtab + e, (((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid]) * R.inv_scale, pre);
}

// 256 elements of the L * 6 * D * D weight gradients (sum of the R.nsplit partials).  vb in [0, nmat_blocks)
template <bool COHERENT = false>
__device__ __forceinline__ void eg_reduce_mat(const EgReduce& R, int vb, int tid) {
    const int64_t e = (int64_t)vb * 256 + tid;
    const int dd = R.D * R.D;
    if (e >= (int64_t)R.L * EG_NMAT * dd) return;
    const int lm = (int)(e / dd), off = (int)(e % dd);
    const int nsplit = R.nsplit;
    const float* p = R.part + (int64_t)lm * nsplit * dd + off;
    const int l = lm / EG_NMAT, m = lm % EG_NMAT;
    float* const* P = R.dst.p[l];
    float* d = (m == 0) ? P[10] : (m == 1) ? P[8] : (m == 2) ? P[4] : P[2] + (m - 3) * dd;
    const EgPre pre = eg_pre(R.AD, d + off);
    float s = 0.f;
    if (nsplit <= 24) {      // (uniform)
        float v[24];
#pragma unroll
        for (int i = 0; i < 24; ++i) v[i] = eg_ld<COHERENT>(p + (int64_t)(i < nsplit ? i : 0) * dd);   // (clamped, unconditional)
#pragma unroll
        for (int i = 0; i < 24; ++i) s += i < nsplit ? v[i] : 0.f;                    // (x + 0 = x: the partials in split order)
    } else {
        float v[WG_NSPLIT_MAX];
#pragma unroll
        for (int i = 0; i < WG_NSPLIT_MAX; ++i) v[i] = eg_ld<COHERENT>(p + (int64_t)(i < nsplit ? i : 0) * dd);
#pragma unroll
        for (int i = 0; i < WG_NSPLIT_MAX; ++i) s += i < nsplit ? v[i] : 0.f;
    }
    eg_put(R.AD, d + off, s, pre);
}

// 64 columns of one (block, vector), summed over the slabs of the workgroups (tiles) that had work: four waves x a fixed order, joined through
// `red` ([4][64] floats of LDS) across ONE barrier the caller provides between the two halves.  job in [0, nvec_blocks) (or beyond: no-op halves)
struct EgVec { float* dvec; EgPre pre; float s; };
__device__ __forceinline__ EgVec eg_reduce_vec_a(const EgReduce& R, int job, int tid, float (*red)[64]) {
    EgVec V{nullptr, EgPre{0.f, 0.f, 0.f, 0.f, 0.f}, 0.f};
    const int lane = tid & 63, wave = tid >> 6;
    if (job >= R.nvec_blocks) { red[wave][lane] = 0.f; return V; }
    const int D = R.D, L = R.L;
    const int cgs = D / 64;
    const int cg = job % cgs, v = (job / cgs) % EG_NVEC, l = job / (cgs * EG_NVEC);
    const EncPlan PL = enc_plan_view(R.planp, R.B, R.S);
    const int n_items = PL.hdr[0];
    // slab rows: one per workgroup that had work, or -- steps that ran one tile per workgroup (enc_tile.hip) -- one per tile
    const int nact = (R.by_tile && PL.hdr[7] == 1) ? PL.hdr[1] : (n_items < R.nwg ? n_items : R.nwg);
    if (!(v >= 10 && l != L - 1)) {
        float* const* P = R.dst.p[l];
        switch (v) {
            case 0: case 1: case 2: V.dvec = P[3] + v * D; break;
            case 3: V.dvec = P[5]; break;
            case 4: V.dvec = P[9]; break;
            case 5: V.dvec = P[11]; break;
            case 6: V.dvec = P[0]; break;
            case 7: V.dvec = P[1]; break;
            case 8: V.dvec = P[6]; break;
            case 9: V.dvec = P[7]; break;
            case 10: V.dvec = P[12]; break;
            default: V.dvec = P[13]; break;
        }
        V.dvec += cg * 64 + lane;
    }
    if (V.dvec && wave == 0) V.pre = eg_pre(R.AD, V.dvec);
    // Round 6: a lane loads FOUR columns (16 bytes) of a slab row and the wave's four lane groups walk four different slabs -- 1 KB per load
    // instruction instead of 256 B.  With one slab per TILE a batch of 4 096 sequences has 1 957 of them and only L x 12 workgroups to sum them:
    // what a workgroup can keep in flight (4 waves x <= 63 loads) set the launch's 24 us there; 16 of these loads are 16 KB per wave.
    // Order (fixed, so the sums are reproducible): lane group r of wave w takes slabs 4 w + r, + 16, + 32, ... in ascending order; then the four
    // groups of a wave pairwise ((r0 + r1) + (r2 + r3)), then the four waves ((w0 + w1) + (w2 + w3)) in eg_reduce_vec_b.
    const int r = lane >> 4, c4 = lane & 15;
    const float* sl = R.slab + ((int64_t)l * EG_NVEC + v) * D + cg * 64 + 4 * c4;
    const int64_t stride = (int64_t)L * EG_NVEC * D;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int EG_Q = 16;
    for (int w0 = 4 * wave + r; w0 < nact; w0 += 16 * EG_Q) {
        float4 x[EG_Q];
#pragma unroll
        for (int q = 0; q < EG_Q; ++q) {
            const int w = w0 + 16 * q;
            x[q] = (w < nact) ? *reinterpret_cast<const float4*>(sl + (int64_t)w * stride) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int q = 0; q < EG_Q; ++q)
            if (w0 + 16 * q < nact) { acc.x += x[q].x; acc.y += x[q].y; acc.z += x[q].z; acc.w += x[q].w; }
    }
    float t[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        t[j] += __shfl_xor(t[j], 16, 64);
        t[j] += __shfl_xor(t[j], 32, 64);
    }
    if (r == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[wave][4 * c4 + j] = t[j];
    }
    return V;
}
__device__ __forceinline__ void eg_reduce_vec_b(const EgReduce& R, const EgVec& V, int tid, float (*red)[64]) {
    const int lane = tid & 63, wave = tid >> 6;
    if (wave != 0 || !V.dvec) return;
    const float s = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    eg_put(R.AD, V.dvec, s, V.pre);
}
