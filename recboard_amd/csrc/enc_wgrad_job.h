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
// The step tail's queue at D = 64 cuts finer from 1 024 sequences on: 40 splits = 240 tickets, one for (nearly) every workgroup of the launch --
// with 144 tickets of a 31 000-row contraction each (B = 4 096) the matrix tickets were 115 k cycles on 144 of the 256 CUs, the longest chain
// of the launch (scripts/tail_phases.py); below that a ticket is three stages and more of them would only queue behind each other.
__host__ __device__ constexpr int wg_nsplit_tail(int D, int64_t B) { return D == 64 && B >= 1024 ? 40 : wg_nsplit(D); }
#define WG_NSPLIT_MAX 40
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
                                              const float* __restrict__ gtape, int64_t NR, int n_tiles, float* __restrict__ part,
                                              int WG_NSPLIT = wg_nsplit(D)) {
    using C = WgCfg<D, NT>;
    static_assert(C::WR >= 1 && C::NS % C::WR == 0, "waves = strips x row groups");
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

// Position-table gradient, partial sums.  The batch's sequences are dealt to WG_POS_GROUPS groups as contiguous ranges; group g writes
// ppart[p][g][D] = sum over ITS sequences (ascending) with a real token at p of contrib[b][p], for every position p.  Only rows of real tokens
// are read (a Beauty-shaped batch is 88 % padding: the round-5 form -- one job per (position, 64 sequences), every row of the dense [B, S, D]
// array loaded and the pads dropped by a select -- was 52 MB of loads and 22 dependent round trips a group at B = 4 096, ~117 k cycles of the
// step tail's ~250 k: scripts/tail_phases.py):
//   1. the range's token flags (the ids' low words != 0) -> LDS, 64 sequences at a time;
//   2. every row group rg lists the (position, sequence) pairs of ITS positions p = rg (mod CG), position-major, sequences ascending (a ballot
//      per position);
//   3. the list's rows, sixteen loads in flight, summed per position in a register and added to acc[p][col] in LDS (one owner per element);
//   4. acc -> ppart.
// The order of every sum is fixed by (B, S) alone; barrier counts as well (enc_tail_k runs two groups side by side behind workgroup barriers).
#define WG_POS_GROUPS 144
template <int D>
__device__ __forceinline__ void wg_pos_job(int tid, float* lds, int g, int B, int S, const int64_t* __restrict__ seq,
                                           const float* __restrict__ contrib, float* __restrict__ ppart) {
    using C = EC<D>;
    constexpr int NT = C::NT, CG = C::CG, G = WG_POS_GROUPS;
    constexpr int PPG = (64 + CG - 1) / CG;                // positions of a row group, at most (S <= 64)
    constexpr int LIST = PPG * 64;                         // its list: 64 sequences a pass
    static_assert((64 * D + 64 * 64 / 4 + CG * LIST / 2) <= wg_job_lds_floats<D>(), "accumulators + flags + lists fit a job's LDS");
    float* acc = lds;                                                                  // [S][D]
    unsigned char* tok = reinterpret_cast<unsigned char*>(lds + 64 * D);               // [64][S]
    unsigned short* list = reinterpret_cast<unsigned short*>(tok + 64 * 64) + (tid / D) * LIST;
    const int nb = (B + G - 1) / G;                        // sequences of a group
    const int b_lo = min(g * nb, B), b_hi = min(b_lo + nb, B);
    const int nsub = (nb + 63) / 64;                       // (uniform: the same for every group)
    const int col = tid % D, rg = tid / D, lane = tid & 63;
    // (an item id's LOW word: ids are < 2^31, and only "is there a token" is asked)
    const int32_t* __restrict__ seq_lo = reinterpret_cast<const int32_t*>(seq);
    __syncthreads();
    for (int i = tid; i < S * D; i += NT) acc[i] = 0.f;
    for (int c = 0; c < nsub; ++c) {
        const int bb = b_lo + 64 * c;
        const int nbc = max(0, min(64, b_hi - bb));
        for (int i = tid; i < nbc * S; i += NT) tok[i] = seq_lo[2 * ((int64_t)bb * S + i)] != 0;      // (tok[bl * S + p]: the batch's own layout)
        __syncthreads();
        int cnt = 0;
        for (int p = rg; p < S; p += CG) {                 // (at D = 128 both waves of a row group write the same list)
            const bool f = lane < nbc && tok[lane * S + p];
            const unsigned long long m = __ballot(f);
            if (f) list[cnt + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned short)(p << 6 | lane);
            cnt += __popcll(m);
        }
        __syncthreads();
        int pcur = -1;
        float a = 0.f;
        for (int k0 = 0; k0 < cnt; k0 += 16) {
            float x[16];
            int pe[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {                 // (clamped, unconditional: the loads of a pass are requested together)
                const int e = __builtin_amdgcn_readfirstlane((int)list[min(k0 + u, cnt - 1)]);
                pe[u] = e >> 6;
                x[u] = contrib[((int64_t)(bb + (e & 63)) * S + pe[u]) * D + col];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (k0 + u < cnt) {                        // (uniform)
                    if (pe[u] != pcur) {
                        if (pcur >= 0) acc[pcur * D + col] += a;
                        pcur = pe[u];
                        a = 0.f;
                    }
                    a += x[u];
                }
            }
        }
        if (pcur >= 0) acc[pcur * D + col] += a;
        __syncthreads();
    }
    for (int i = tid; i < S * D; i += NT) ppart[((int64_t)(i / D) * G + g) * D + i % D] = acc[i];
}
