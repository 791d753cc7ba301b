// Weight preparation of the one-tile-per-workgroup SASRec step (enc_tile.hip): constants, the bf16 hi / mid split, and the per-thread body
// that turns the fp32 [64][64] matrices into fragment planes.  Shared by enc_tile.hip (enc_tile_prep_k) and enc_plan.hip (the batch
// preparation launch does it in extra workgroups when asked: re_sasrec_batch_prep_w).
#pragma once
#include "enc_common.h"

typedef __bf16 tl_bf16x8 __attribute__((ext_vector_type(8)));
typedef short tl_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned tl_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned tl_u32x2 __attribute__((ext_vector_type(2)));

// sizes as functions of ns = strips of 16 features (D = 16 ns: 4 at D = 64, 8 at D = 128)
#define TL_NPAR 10              // per block: 0 ln_a_w 1 ln_a_b 2 bq 3 bk 4 bv 5 bo 6 ln_f_w 7 ln_f_b 8 b1 9 b2
#define TLC_D(ns) (16 * (ns))
#define TLC_NQ(ns) ((ns) / 2)
#define TLC_FRAG_WORDS(ns) ((ns) * TLC_NQ(ns) * 2 * 64 * 4)   // one (block, matrix, orientation): [strip][k step][plane 2][lane 64] x 16 bytes
// per tile, in floats: operand slots [3][2 planes][64 lanes][strips] x 8 B | partial score tiles | per-token partials [2][waves][16 tokens] x 16 B |
// transpose scratch [waves][16 x 20]
#define TLC_OB(ns) (3 * 2 * 64 * (ns) * 2)
#define TLC_RED(ns) (2 * (ns) * 64 * 16)   // partial score tiles [2][waves][64 lanes][4 key tiles] x 16 B
#define TLC_SM(ns) (2 * (ns) * 16 * 4)
#define TLC_TR(ns) ((ns) * 320)
#define TLC_TILE_LDS(ns) (TLC_OB(ns) + TLC_RED(ns) + TLC_SM(ns) + TLC_TR(ns))
// floats of a key tile's inbox of partial dV / dK for ONE block: [sender t - kt - 1][dV, dK][strip][4 registers][64 lanes].  A tile has an inbox PER
// BLOCK ([tile][l]): the last tile of a sequence waits for nobody in the backward pass, so it can be a whole block ahead of the tile it sends
// to -- with one inbox for all blocks its block l - 1 partial could land before the receiver had added the block l one (round 4's finding:
// results that differed from run to run whenever co-resident workgroups slowed some tiles of a sequence down)
#define TLC_XCH_TILE(ns) (3 * 2 * (ns) * 256)
#define TLC_PREP_THREADS(L, ns) (6 * (L) * 2 * (ns) * TLC_NQ(ns) * 64)

inline bool enc_tile_width_ok(int64_t D) { return D == 64 || D == 128; }
// (+ the small parameters of every block and lastLN as one block of (10 L + 2) x D floats behind the fragments)
inline size_t enc_tile_wfrag_bytes(int64_t L, int64_t D) { return (size_t)L * 6 * 2 * TLC_FRAG_WORDS(D / 16) * 4 + (size_t)(TL_NPAR * L + 2) * D * 4; }
inline size_t enc_tile_xch_bytes(int64_t B, int64_t S, int64_t D, int64_t L) {
    const int64_t mt = enc_plan_max_tiles(B, S);
    return (size_t)(mt < ENC_XCH_TILE_CAP ? mt : ENC_XCH_TILE_CAP) * L * TLC_XCH_TILE(D / 16) * 4;
}
// where the fragments, the exchange inboxes and the launch epoch live: behind the gradient tape in the backward workspace / in the tape's flag area
inline uint32_t* enc_tile_wf(float* gtape, int64_t B, int64_t S, int64_t L, int64_t D) {
    return (uint32_t*)((((uintptr_t)(gtape + (size_t)L * EG_NMAT * 16 * enc_plan_max_tiles(B, S) * D)) + 255) & ~(uintptr_t)255);
}
inline float* enc_tile_xch(uint32_t* wf, int64_t L, int64_t D) { return (float*)(((uintptr_t)wf + enc_tile_wfrag_bytes(L, D) + 255) & ~(uintptr_t)255); }
// The backward workspace (re_sasrec_encoder_bwd_workspace_bytes), carved in ONE place: workgroup slabs of vector gradients | the matrices' split-K
// partials | the position table's group partials | the gradient tape | (256-byte aligned) the tile kernels' weight fragments + small parameters |
// (aligned) the tiles' dK / dV inboxes.  `bytes` = the first byte behind the last region, from `ws`.  Every entry point that touches the workspace
// goes through this (enc_bwd.hip, enc_step.hip, enc_tail.hip, enc_plan.hip); tests/test_workspace_layout.py checks it against the size query and
// against the largest index each kernel can form (re_sasrec_encoder_bwd_workspace_layout).
size_t enc_wgrad_part_floats(int64_t D, int64_t L);
size_t enc_wgrad_ppart_floats(int64_t B, int64_t D);
struct EncBwdWs {
    float *slab, *wpart, *ppart, *gtape;
    uint32_t* wf;
    float* xch;
    size_t bytes;
};
inline EncBwdWs enc_bwd_ws(void* ws, int64_t B, int64_t S, int64_t D, int64_t L) {
    EncBwdWs W;
    W.slab = (float*)ws;
    W.wpart = W.slab + (size_t)enc_slab_rows(B, S) * L * EG_NVEC * D;
    W.ppart = W.wpart + enc_wgrad_part_floats(D, L);
    W.gtape = W.ppart + enc_wgrad_ppart_floats(B, D);
    W.wf = enc_tile_wf(W.gtape, B, S, L, D);
    W.xch = enc_tile_xch(W.wf, L, D);
    W.bytes = (size_t)((char*)W.xch + enc_tile_xch_bytes(B, S, D, L) - (char*)ws);
    return W;
}
inline unsigned* enc_tile_epoch(void* tape, int64_t B, int64_t S, int64_t L, int64_t D) {
    const EncTape T = enc_tape_layout(B, S, D, L);
    return reinterpret_cast<unsigned*>((float*)tape + T.off_FLAGS) + enc_plan_max_tiles(B, S) * EP_FLAG_WORDS + 1;
}
__host__ __device__ inline size_t tl_lds_floats(int L, int ns) { return (size_t)(TL_NPAR * L + 2) * 16 * ns + (size_t)TLC_TILE_LDS(ns) + 32; }


// ---- operand splits
__device__ __forceinline__ unsigned tl_pk(float a, float b) {   // two fp32 -> packed bf16 (round to nearest even), a in the low half
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ void tl_split2(float a, float b, unsigned& h, unsigned& m) {
    h = tl_pk(a, b);
    m = tl_pk(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xFFFF0000u));   // (x - hi is exact)
}

// ---- weight preparation: fp32 [64][64] -> bf16 hi / mid fragment planes in the kernel's k order, both orientations ---------------------
// thread t of TLC_PREP_THREADS(L, NS): called by enc_tile_prep_k and by the batch preparation kernel's extra workgroups
template <int NS>
__device__ __forceinline__ void tl_prep_thread(const SasrecParams& P, int L, uint32_t* __restrict__ wf, unsigned* __restrict__ epoch, int t) {
    constexpr int TL_D = 16 * NS, NQ = NS / 2, LQ = NS == 4 ? 1 : 2, LS = NS == 4 ? 2 : 3;
    constexpr int TL_FRAG_WORDS = TLC_FRAG_WORDS(NS);
    if (t == 0) {
        epoch[0] += 1u;   // the launch's epoch: what this step's hand-over flags are set to (the step kernel runs behind this one)
        epoch[1] = 0u;    // ... and its counter of tiles handed out beyond the grid (enc_tile_body.inc: the word behind the epoch)
    }
    if (t < (TL_NPAR * L + 2) * TL_D) {   // the small parameters, gathered into one block (the step kernel then needs no parameter table)
        const int v = t / TL_D, cc = t % TL_D;
        const float* p;
        if (v >= TL_NPAR * L) p = (v == TL_NPAR * L) ? P.last_w : P.last_b;
        else {
            const SasrecBlockParams& Wv = P.blk[v / TL_NPAR];
            const int kk = v % TL_NPAR;
            p = (kk == 0) ? Wv.ln_a_w : (kk == 1) ? Wv.ln_a_b : (kk < 5) ? Wv.in_b + (kk - 2) * TL_D : (kk == 5) ? Wv.out_b : (kk == 6) ? Wv.ln_f_w
              : (kk == 7) ? Wv.ln_f_b : (kk == 8) ? Wv.b1 : Wv.b2;
        }
        reinterpret_cast<float*>(wf + (size_t)L * 6 * 2 * TL_FRAG_WORDS)[t] = p[cc];
    }
    const int lane = t & 63, q = (t >> 6) & (NQ - 1), s = (t >> (6 + LQ)) & (NS - 1), o = (t >> (6 + LQ + LS)) & 1, lm = t >> (7 + LQ + LS);
    if (lm >= 6 * L) return;
    const int l = lm / 6, m = lm % 6, c = lane & 15, g = lane >> 4;
    const SasrecBlockParams& W = P.blk[l];
    const float* w = (m < 3) ? W.in_w + m * TL_D * TL_D : (m == 3) ? W.out_w : (m == 4) ? W.w1 : W.w2;
    unsigned h[4], md[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * p + e;
            const int k = 16 * (2 * q + (i >> 2)) + 4 * g + (i & 3);
            v[e] = (o == 0) ? w[(16 * s + c) * TL_D + k] : w[k * TL_D + 16 * s + c];   // o = 0: y = x W^T, o = 1: dx = dy W
        }
        tl_split2(v[0], v[1], h[p], md[p]);
    }
    tl_u32x4* dst = reinterpret_cast<tl_u32x4*>(wf + ((size_t)lm * 2 + o) * TL_FRAG_WORDS) + lane;
    dst[((s * NQ + q) * 2 + 0) * 64] = (tl_u32x4){h[0], h[1], h[2], h[3]};
    dst[((s * NQ + q) * 2 + 1) * 64] = (tl_u32x4){md[0], md[1], md[2], md[3]};
}

