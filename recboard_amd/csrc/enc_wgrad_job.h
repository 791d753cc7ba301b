// The two kinds of jobs of the encoder's weight-gradient launch, as device functions of a GROUP of EC<D>::NT = 512 threads: enc_wgrad_k
// (enc_wgrad.hip) runs one job per workgroup; enc_tail_k (enc_tail.hip) runs them two at a time in the halves of the 1024-thread workgroups
// that have finished the item table's scatter-add.  Every barrier is workgroup-wide, so a job's barrier count depends on the launch's shapes
// only (never on the job): groups that run different jobs of one kind side by side stay in step.
#pragma once
#include "enc_common.h"
#include "enc_tile_prep.h"

// row splits of a matrix's contraction (partials per matrix): 24 at D = 64; 12 at D = 128 -- there a job of the 1024-thread tail launches is
// run by the WHOLE workgroup (sixteen waves, 4 accumulator tiles and one staging unit a thread instead of 8 and 2): two 512-thread jobs side by
// side, as at D = 64, need more than the 128 registers such a thread has and spilled into scratch memory inside the stage loop (a stage 9 - 14 k
// cycles instead of ~4 k: config 5's tail 47 us, scripts/tail_phases.py large).  Half as many splits of twice the rows keep the ticket count.
__host__ __device__ constexpr int wg_nsplit(int D) { return D == 128 ? 12 : 24; }
#define WG_NSPLIT_MAX 24
#define WG_CH 4   // row tiles per LDS stage
#ifndef WG_STAMP
#define WG_STAMP(i) do { } while (0)   // (enc_tail.hip's profile build: shader-clock stamps inside a job)
#endif

template <int D>
__host__ __device__ constexpr int wg_job_lds_floats() { return 2 * 16 * WG_CH * EC<D>::LS; }

// dW partial of (block l, matrix m, row split): out[D][D] = sum over the split's row tiles of dY^T X.   tid: 0 .. 511 within the group.
//   gradient-tape order: 0 dO2 (x HR -> W2)  1 dH (x Y -> W1)  2 dX1 (x O -> Wo)  3 dQ (x A -> Wq)  4 dK (x X -> Wk)  5 dV (x X -> Wv)
// The contraction runs over the batch's ROWS, on the XDL pipe as bf16 hi / mid three-product splits (the tile kernels' arithmetic:
// hi = bf16(x), mid = bf16(x - hi); hi*hi + hi*mid + mid*hi, fp32 accumulation; <= 3.01 * 2^-18 relative per product): sixteen waves of fp32
// MFMAs per CU were what a job took (7.7 of the ~8 us a stage took at D = 128).  Both operands are split where they are STAGED -- once per
// element, not once per wave that reads it: a thread fetches the same four columns of two consecutive rows and writes them as four
// {row 2p, row 2p + 1} bf16 pairs per plane, so that a lane's MFMA operand (eight consecutive rows of one column) is four words of a column.
template <int D, int NT>
struct WgCfg {
    static constexpr int NS = EC<D>::NS, WR = (NT / 64) / EC<D>::NS;   // column strips, wave row groups
};
// NT: threads of the group that runs the job (tid: 0 .. NT - 1): 512, or 1024 (enc_tail.hip at D = 128)
template <int D, int NT = EC<D>::NT>
__device__ __forceinline__ void wg_matrix_job(int tid, float* lds, int l, int m, int split, const float* __restrict__ tape, const EncTape& T,
                                              const float* __restrict__ gtape, int64_t NR, int n_tiles, float* __restrict__ part) {
    using C = WgCfg<D, NT>;
    static_assert(C::WR >= 1 && C::NS % C::WR == 0, "waves = strips x row groups");
    constexpr int WG_NSPLIT = wg_nsplit(D);
    constexpr int RTW = C::NS / C::WR;   // output row tiles per wave
    constexpr int RSW = D + 4;           // words of a row pair (+ 4: the four lane groups of a fragment read land in four bank quarters)
    constexpr int NPAIR = 16 * WG_CH / 2;
    constexpr int PLANE = NPAIR * RSW;
    static_assert(4 * PLANE == wg_job_lds_floats<D>(), "four planes (dY hi, mid, X hi, mid) fill the job's LDS");
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int strip = wave % C::NS, wr = wave / C::NS, g = lane >> 4, c = lane & 15;
    const int per = (n_tiles + WG_NSPLIT - 1) / WG_NSPLIT;
    const int trips = (per + WG_CH - 1) / WG_CH;      // (the same for every split: the last ones run short or empty stages)
    const int t0 = split * per;
    const int t1 = (t0 + per < n_tiles) ? t0 + per : n_tiles;
    const int64_t xoff = (m == 0) ? T.off_HR : (m == 1) ? T.off_Y : (m == 2) ? T.off_O : (m == 3) ? T.off_A : T.off_X;
    const float* X = tape + (int64_t)l * T.per_block + xoff;
    const float* dY = gtape + ((int64_t)l * EG_NMAT + m) * NR * D;
    uint32_t* aH = reinterpret_cast<uint32_t*>(lds);
    uint32_t* aM = aH + PLANE;
    uint32_t* bH = aM + PLANE;
    uint32_t* bM = bH + PLANE;
    f32x4 acc[RTW];
#pragma unroll
    for (int t = 0; t < RTW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Stages of WG_CH row tiles through LDS.  Every stage's rows are a memory round trip (~2 us under the launch's own load) and a job has
    // ~3 of them: PF stages are requested into registers at once, then staged and multiplied one after the other, a slot's next stage being
    // requested as soon as its registers are free.  PF = 1 (the next stage in flight under this stage's products) is what the registers of
    // enc_tail_k allow: PF = 3 at D = 64 -- a Beauty-shaped job as ONE round trip -- pushed 36 registers of the 128 a thread of a
    // 1024-thread workgroup has into scratch memory, and a job took 38 k cycles instead of 24 k (scripts/tail_phases.py, round 4).
    constexpr int NP = NPAIR * (D / 4) / NT;      // (row pair, four columns) units per thread: 1 at D = 64, 2 at D = 128 (1 with 1024 threads)
    constexpr int PF = 1;
    f32x4 ra[PF][NP][2], rb[PF][NP][2];   // (native vectors and a macro: HIP's float4 struct arrays / arrays captured by a lambda stay in scratch memory)
#define WG_FETCH(Q, TC)                                                                                         \
    do {                                                                                                        \
        const int ntc_ = (t1 - (TC)) < WG_CH ? (t1 - (TC)) : WG_CH;                                             \
        const int nf_ = ntc_ > 0 ? 16 * ntc_ * (D / 4) : 1;                                                     \
        const int tc_ = ntc_ > 0 ? (TC) : t0;   /* (a stage behind the split's rows: any valid address, its values are not used) */ \
        _Pragma("unroll") for (int j = 0; j < NP; ++j) {                                                        \
            const int u_ = j * NT + tid, p_ = u_ / (D / 4), c4_ = u_ % (D / 4);                              \
            _Pragma("unroll") for (int e = 0; e < 2; ++e) {                                                     \
                int f = (2 * p_ + e) * (D / 4) + c4_;                                                           \
                f = f < nf_ ? f : nf_ - 1;   /* clamped, unconditional: a predicated load is waited for on the spot */ \
                ra[Q][j][e] = reinterpret_cast<const f32x4*>(dY + (int64_t)tc_ * 16 * D)[f];                    \
                rb[Q][j][e] = reinterpret_cast<const f32x4*>(X + (int64_t)tc_ * 16 * D)[f];                     \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)
    WG_STAMP(0);
    if (t0 < t1) {
#pragma unroll
        for (int q = 0; q < PF; ++q) WG_FETCH(q, t0 + q * WG_CH);
    }
    WG_STAMP(1);
    for (int it0 = 0; it0 < trips; it0 += PF) {
#pragma unroll
        for (int q = 0; q < PF; ++q) {
            const int it = it0 + q;
            if (it >= trips) break;                  // (uniform)
            const int tc = t0 + it * WG_CH;
            const int left = t1 - tc;
            const int ntc = left < WG_CH ? (left > 0 ? left : 0) : WG_CH;
            enc_sync();
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int u = j * NT + tid, p = u / (D / 4), c4 = u % (D / 4);
                const bool live = 2 * p < 16 * ntc;      // (rows come in pairs: 16 ntc is even; pairs behind the stage's rows are zero)
                tl_u32x4 h, md, h2, md2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    unsigned x, y;
                    tl_split2(live ? ra[q][j][0][i] : 0.f, live ? ra[q][j][1][i] : 0.f, x, y);
                    h[i] = x; md[i] = y;
                    tl_split2(live ? rb[q][j][0][i] : 0.f, live ? rb[q][j][1][i] : 0.f, x, y);
                    h2[i] = x; md2[i] = y;
                }
                *reinterpret_cast<tl_u32x4*>(aH + p * RSW + 4 * c4) = h;
                *reinterpret_cast<tl_u32x4*>(aM + p * RSW + 4 * c4) = md;
                *reinterpret_cast<tl_u32x4*>(bH + p * RSW + 4 * c4) = h2;
                *reinterpret_cast<tl_u32x4*>(bM + p * RSW + 4 * c4) = md2;
            }
            // (this slot's registers are free again: the stage PF further on -- none on a Beauty-shaped batch at D = 64)
            if (tc + PF * WG_CH < t1) WG_FETCH(q, tc + PF * WG_CH);
            enc_sync();
            if (it == 0) WG_STAMP(2);
            const int ksteps = (ntc + 1) >> 1;           // 32 rows (two row tiles) per MFMA
            for (int ks = 0; ks < ksteps; ++ks) {
                const int w0 = (16 * ks + 4 * g) * RSW + c;       // the lane's four row pairs: rows 32 ks + 8 g + (0 .. 7)
                tl_u32x4 bh, bm;
#pragma unroll
                for (int i = 0; i < 4; ++i) { bh[i] = bH[w0 + i * RSW + 16 * strip]; bm[i] = bM[w0 + i * RSW + 16 * strip]; }
#pragma unroll
                for (int t = 0; t < RTW; ++t) {
                    const int mt = t * C::WR + wr;
                    tl_u32x4 ah, am;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { ah[i] = aH[w0 + i * RSW + 16 * mt]; am[i] = aM[w0 + i * RSW + 16 * mt]; }
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, am), __builtin_bit_cast(tl_bf16x8, bh), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, ah), __builtin_bit_cast(tl_bf16x8, bm), acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, ah), __builtin_bit_cast(tl_bf16x8, bh), acc[t], 0, 0, 0);
                }
            }
            if (it < 3) WG_STAMP(3 + it);
        }
    }
#undef WG_FETCH
    float* out = part + (((int64_t)l * EG_NMAT + m) * WG_NSPLIT + split) * D * D;
#pragma unroll
    for (int t = 0; t < RTW; ++t) {
        const int mt = t * C::WR + wr;
#pragma unroll
        for (int j = 0; j < 4; ++j) out[(16 * mt + 4 * g + j) * D + 16 * strip + c] = acc[t][j];
    }
}

// Position-table gradient, partial sums: job (p, chunk of 64 sequences) -> ppart[p][chunk][D] = sum over the chunk's sequences with a real
// token at p of contrib[b][p]  (every load independent: one memory round trip per job).  This group takes jobs j0, j0 + jstep, ...
template <int D>
__device__ __forceinline__ void wg_pos_job(int tid, float* lds, int j0, int jstep, int B, int S, const int64_t* __restrict__ seq,
                                           const float* __restrict__ contrib, float* __restrict__ ppart) {
    using C = EC<D>;
    const int nch = (B + 63) / 64, njobs = S * nch;
    const int trips = (njobs + jstep - 1) / jstep;    // (the same for every j0)
    const int col = tid % D, rg = tid / D;
    // (the next job's loads are in flight while this one is reduced: a group has ~3 jobs, each a memory round trip)
    constexpr int NQP = 64 / C::CG;
    // (an item id's LOW word: ids are < 2^31, and only "is there a token" is asked -- half the registers of the 64-bit ids, which at D = 128,
    //  16 rows a thread and two jobs in flight, did not fit the tail launches' 128)
    const int32_t* __restrict__ seq_lo = reinterpret_cast<const int32_t*>(seq);
    int32_t sv[NQP], svn[NQP];
    float cv[NQP], cvn[NQP];
#define WG_PJOB(J, SV, CV)                                                              \
    do {                                                                                \
        const int p_ = (J) / nch, ch_ = (J) % nch;                                      \
        _Pragma("unroll") for (int q = 0; q < NQP; ++q) {                               \
            int b = ch_ * 64 + rg + C::CG * q;                                          \
            const bool in_ = b < B;                                                     \
            b = in_ ? b : B - 1;   /* clamped, unconditional loads */                   \
            const int32_t sx = seq_lo[2 * ((int64_t)b * S + p_)];                       \
            const float cx = contrib[((int64_t)b * S + p_) * D + col];                  \
            SV[q] = in_ ? sx : 0;                                                       \
            CV[q] = cx;                                                                 \
        }                                                                               \
    } while (0)
#pragma unroll
    for (int q = 0; q < NQP; ++q) { sv[q] = 0; cv[q] = 0.f; svn[q] = 0; cvn[q] = 0.f; }
    if (j0 < njobs) WG_PJOB(j0, sv, cv);
    for (int it = 0; it < trips; ++it) {
        const int j = j0 + it * jstep;
        if (j + jstep < njobs) WG_PJOB(j + jstep, svn, cvn);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NQP; ++q) s += (sv[q] != 0) ? cv[q] : 0.f;   // (rows of pad positions are never written: select, not multiply)
        __syncthreads();
        lds[tid] = s;
        __syncthreads();
        if (tid < D && j < njobs) {
            float t = lds[tid];
#pragma unroll
            for (int i = 1; i < C::CG; ++i) t += lds[i * D + tid];
            ppart[(int64_t)j * D + tid] = t;
        }
#pragma unroll
        for (int q = 0; q < NQP; ++q) { sv[q] = svn[q]; cv[q] = cvn[q]; }
    }
#undef WG_PJOB
}
