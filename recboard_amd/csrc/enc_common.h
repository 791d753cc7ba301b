// Shared device code of the fused SASRec encoder kernels (enc_fwd.hip, enc_bwd.hip, enc_wgrad.hip, enc_plan.hip).
//
// Reference restated: SASRec/main.py:163-176 (after_one_block), :31-50 (PointWiseFeedForward), :178-193 (encode).
//
// WORK ITEMS.  Sequences are left-padded (SASRec/main.py:143-157: lpad_), so a sequence's real tokens are its last `span`
// positions and the `first = S - span` positions in front of them are pads.  Pad positions are attended as keys by the
// reference, but they are all the SAME key (x = 0 -> k = b_k, v = b_v): they enter the softmax analytically as one virtual
// key of multiplicity `first`, forward and backward, and get no rows.  Only the rows from the first real token on are
// materialised, in 16-row TILES (one MFMA tile): sequences of span <= 16 share tiles (power-of-two slots, sorted by size:
// no slot straddles a tile; attention is restricted to same-sequence rows), a longer sequence owns ceil(span/16) tiles.
// A work item = 1..MAXT consecutive tiles: one long sequence, or G tiles of short ones (G chosen by the plan kernel so that
// the items just fill the chip).  On Beauty-shaped batches (88 % of the token slots are padding) that is ~3 600 rows in
// ~230 tiles instead of 25 600 token slots, spread over every CU.  `re_sasrec_batch_prep` (enc_plan.hip) builds the plan
// on the device (no host sync, capturable).
//
// One workgroup (8 waves) per item; all of an item's activations live in LDS as [ROWS][D + 4] fp32 tiles; every product is
// v_mfma_f32_16x16x4_f32 (exact fp32).  Wave (wr, strip) owns output columns [16 strip, +16) of the row tiles
// tt = t * WR + wr.  The k index of an MFMA step is free as long as A and B agree: lane group g = lane >> 4 takes
// k = 16 q + 4 g + i at step s = 4 q + i, so a k-contiguous operand is read as one 16-byte load per 16 of k, and a
// k-strided one (rows 16 q + 4 g + i) hits disjoint LDS banks in the two lane groups of a ds_read_b32.
#pragma once
#include "re_common.h"
#include "re_rng.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

// A value the compiler must treat as freshly produced here.  The GEMM helpers' epilogue rows go through it: their LDS addresses
// (row * LS + col for four rows) were otherwise computed once outside the block loop, kept alive across it, spilled to scratch, and each
// reload's `s_waitcnt vmcnt(0)` -- the memory counter retires in order -- drained the tile prefetches in flight (a full memory round
// trip in the middle of a phase).  Recomputed at the point of use they cost two vector instructions and the four rows share one register.
__device__ __forceinline__ int enc_opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <int D>
struct EC {
    static constexpr int NS = D / 16;                 // column strips
    static constexpr int NW = 8;                      // waves per workgroup
    static constexpr int NT = 64 * NW;                // threads
    static constexpr int WR = NW / NS;                // wave row groups (2 at D = 64, 1 at D = 128)
    static constexpr int MAXT = (D == 64) ? 4 : 2;    // tiles per work item (LDS capacity)
    static constexpr int RT = MAXT / WR;              // row tiles per wave
    static constexpr int ROWS = 16 * MAXT;
    static constexpr int TPR = NT / ROWS;             // row-wise phases: threads per row (8 / 16) ...
    static constexpr int CPT = D / TPR;               // ... columns per thread (8)
    static constexpr int LS = D + 4;                  // LDS row stride (floats): conflict-free 16-byte row reads
    static constexpr int PLS = 64 + 4;                // row stride of the [ROWS][64] probability tiles (a sequence has at most 64 keys)
    static constexpr int KPT = 64 / TPR;              // softmax phases: keys per thread
    static constexpr int BUF = ROWS * LS;
    static constexpr int PRE = 32 * LS;               // a prefix key tile pair (the first half of a split / chained sequence: 32 rows)
    static constexpr int PBUF = ROWS * PLS;
    static constexpr int KS = D / 4;                  // MFMA steps of a contraction over D
    static constexpr int CG = NT / D;                 // column-sum phases: row groups ...
    static constexpr int RPW = ROWS / CG;             // ... rows per thread (8)
};

struct SasrecBlockParams {
    const float *ln_a_w, *ln_a_b;   // attnLNs.l
    const float *in_w, *in_b;       // attnLayers.l.in_proj_{weight,bias}  [3D, D], [3D]
    const float *out_w, *out_b;     // attnLayers.l.out_proj
    const float *ln_f_w, *ln_f_b;   // fwdLNs.l
    const float *w1, *b1, *w2, *b2; // fwdLayers.l.conv{1,2} ([D, D, 1] == [D, D])
};
#define SE_MAX_BLOCKS 4
struct SasrecParams {
    SasrecBlockParams blk[SE_MAX_BLOCKS];
    const float *last_w, *last_b;
};

// ---- plan (re_sasrec_batch_prep): int32 words
//   [0] n_items  [1] n_tiles  [2] n_long items  [3] tiles per short item  [4] number of valid (non-pad) positions
//   [5] number of sequences SPLIT over two work items (kinds 2 / 3; 0 = none)  [6] the workgroup count the plan was made for
//   [7] 1 = every tile can have a resident workgroup of its own (the one-tile-per-workgroup step of enc_tile.hip may run)
//   [8 .. 8 + MT)            item descriptors: tile0 | nt << 24 | kind << 28   (kind 1 = one sequence over nt tiles)
//   then int2 rowmap[MT * 16]: { gid = b * S + s or -1 (dummy row), first = pads in front of the row's sequence }
//   then scratch of the plan kernel.   MT = B * ceil(S / 16) bounds the number of tiles.
#define EP_HDR 8
#define EP_PW 64   // row width of the saved probabilities: a sequence has at most 64 keys
#define EP_FLAG_WORDS 8   // per tile: [l] forward k, v of block l published, [4 + l] backward dK, dV partials of block l published
struct EncPlan {
    const int* hdr;
    const int* items;
    const int2* rowmap;
};
__host__ __device__ inline int64_t enc_plan_max_tiles(int64_t B, int64_t S) { return B * ((S + 15) / 16); }
// which form of the tile kernels a launch of this shape uses (enc_tile_body.inc: enc_tile_step_k<LOOP>): batches of more than 2048 possible
// tiles (B > 512 at S = 50) the looped one -- the plan (enc_plan_body.h) applies the matching rule
#ifndef ENC_TILE_LOOP_FROM
#define ENC_TILE_LOOP_FROM 1024
#endif
__host__ __device__ inline bool enc_tile_looped(int64_t B, int64_t S) { return enc_plan_max_tiles(B, S) > ENC_TILE_LOOP_FROM; }
// resident workgroups per CU of the tile kernels (enc_tile.hip).  TWO at D = 64 since round 6 (four waves, ~206 registers, 58 KB of LDS each: 14 - 20 %
// faster on batches of 1 024 - 4 096 sequences); ONE at D = 128 (eight waves, 234 registers).  Rounds 3 - 5 shipped one: with two the results differed
// from replay to replay -- the LOW register of a packed-fp32 result (v_pk_mul / add / fma_f32) wrong in its last sixteen lanes whenever two waves
// shared a SIMD.  enc_tile.hip is compiled without packed-fp32 instructions now (Makefile: TILE_FLAGS; profiles/r6_handover_notes.txt), which also
// covers the D = 128 kernels, whose eight waves put two on a SIMD at ONE workgroup per CU.  The launcher, the plan's rule (split_long & 8: the host
// asks re_tile_wgs_per_cu) and the looped grid all follow this one number.
// the most tiles a batch may have for the tile kernels to run it (the plan's rule, enc_plan_body.h): their dK / dV inboxes (enc_tile_prep.h:
// enc_tile_xch_bytes) are sized for this many tiles, not for every tile a batch of B sequences could have (B = 8 192: 32 768 tiles = 1.6 GB of
// inboxes that the plan's speed rule -- at most ~10 tiles per resident workgroup -- never let the kernels use)
#define ENC_XCH_TILE_CAP 5120
#ifndef ENC_TILE_WG_PER_CU
#define ENC_TILE_WG_PER_CU 2      // (D = 64; `make one` builds the one-per-CU library for A/B runs)
#endif
__host__ __device__ inline int enc_tile_wg_per_cu(int64_t D) { return D == 64 ? ENC_TILE_WG_PER_CU : 1; }
// rows of the vector-gradient slab in the backward's workspace: one per workgroup (<= 1024) or one per tile (enc_tile.hip)
__host__ __device__ inline int64_t enc_slab_rows(int64_t B, int64_t S) { const int64_t mt = enc_plan_max_tiles(B, S); return mt > 1024 ? mt : 1024; }
__host__ __device__ inline int64_t enc_plan_rowmap_word(int64_t B, int64_t S) { return (EP_HDR + enc_plan_max_tiles(B, S) + 1) / 2 * 2; }
__host__ __device__ inline size_t enc_plan_bytes(int64_t B, int64_t S) {
    const int64_t mt = enc_plan_max_tiles(B, S);
    return (size_t)(enc_plan_rowmap_word(B, S) + 2 * 16 * mt + 2 * B + 64) * 4;   // + span / placement scratch
}
__host__ __device__ inline EncPlan enc_plan_view(const void* plan, int64_t B, int64_t S) {
    const int* w = (const int*)plan;
    return EncPlan{w, w + EP_HDR, (const int2*)(w + enc_plan_rowmap_word(B, S))};
}

// ---- tape (activations saved by the forward for the backward), fp32, indexed by COMPACT row (tile * 16 + r): an item's
// rows are one contiguous block of every array.  Per block l:
//   X, A = LN_a(x), Q, K, V, O, X1, Y = LN_f(x1), HR : [NR][D]     P : [NR][ROWS] (pre-dropout probabilities, item-local keys)
//   SA, SF : [NR][2] (mean, rstd)   PP : [NR][2] (probability of one virtual pad key, total kept weight)   MK : [NR][D / 4] mask words (enc_tile.hip)
// then XL [NR][D] (input of lastLN) and SL [NR][2].  NR = 16 * MT.
struct EncTape {
    int64_t per_block, off_X, off_A, off_Q, off_K, off_V, off_O, off_X1, off_Y, off_HR, off_P, off_SA, off_SF, off_PP, off_MK, off_XL, off_SL, off_FLAGS, total;
};
__host__ __device__ inline EncTape enc_tape_layout(int64_t B, int64_t S, int64_t D, int64_t L) {
    EncTape t;
    const int64_t nr = 16 * enc_plan_max_tiles(B, S), act = nr * D, rows = EP_PW;
    int64_t o = 0;
    t.off_X = o; o += act;
    t.off_A = o; o += act;
    t.off_Q = o; o += act;
    t.off_K = o; o += act;
    t.off_V = o; o += act;
    t.off_O = o; o += act;
    t.off_X1 = o; o += act;
    t.off_Y = o; o += act;
    t.off_HR = o; o += act;
    t.off_P = o; o += nr * rows;
    t.off_SA = o; o += nr * 2;
    t.off_SF = o; o += nr * 2;
    t.off_PP = o; o += nr * 2;
    t.off_MK = o; o += nr * (D / 4);                                  // dropout / relu mask words of the one-tile-per-workgroup step (enc_tile.hip): [tile][D / 16 strips][64]
    t.per_block = o;
    t.off_XL = L * o;
    t.off_SL = t.off_XL + act;
    t.off_FLAGS = t.off_SL + nr * 2;                                  // hand-over flags of split sequences (EP_FLAG_WORDS per tile) + one error word
    t.total = t.off_FLAGS + enc_plan_max_tiles(B, S) * EP_FLAG_WORDS + 16;
    return t;
}
// ---- gradient tape written by the backward for the weight-gradient kernel: per block six [NR][D] arrays
//   0 dO2 (-> dW2 with HR)  1 dH (-> dW1 with Y)  2 dX1 (-> dWo with O)  3 dQ (-> dWq with A)  4 dK (-> dWk with X)  5 dV (-> dWv with X)
#define EG_NMAT 6
// vector gradients per block: 0 bq 1 bk 2 bv 3 bo 4 b1 5 b2 6 ga 7 ba 8 gf 9 bf 10 glast 11 blast
#define EG_NVEC 12

// Workgroup barrier for LDS hand-offs.  Nothing that crosses waves inside a phase of these kernels goes through global memory, so the
// barrier only has to order LDS traffic (where global memory does cross waves or workgroups -- tape rows read back by other threads, flags
// behind published rows -- the sites use re_sync_full(): a vmcnt(0) drain in front of the barrier; `__syncthreads()` does NOT include
// one on gfx950, re_common.h): the wave's own LDS operations are complete
// (lgkmcnt(0)), the memory clobber keeps the compiler from moving accesses across it, and vector-memory operations stay in flight.
__device__ __forceinline__ void enc_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Phase stamps (diagnostic build only: `make -C recboard_amd/csrc encprof` -> librecengine_encprof.so, scripts/enc_phases.py):
// thread 0 of workgroup 0 records the shader clock at phase boundaries of its first work item (the plan's largest).
#ifdef ENC_PROFILE
#define ENC_MARKS 96
#define ENC_MARK(arr, i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && k == 0 && (i) < ENC_MARKS) arr[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define ENC_MARK(arr, i) do { } while (0)
#endif

// A pointer whose provenance the compiler no longer knows: loads through it are ordinary loads issued where they are written.
// (Loads from a `const __restrict__` kernel argument are invariant: the compiler re-issues them right in front of their first
// use -- a full L2 round trip inside a product -- instead of keeping registers live across the phases in between.)  The
// explicit global address space keeps them global_load (a generic pointer would turn them into flat_load, which counts on
// both memory counters and retires out of order).
typedef const __attribute__((address_space(1))) float* gcf_t;
__device__ __forceinline__ gcf_t g_launder(const float* p) {
    asm volatile("" : "+s"(p));
    return (gcf_t)p;
}
__device__ __forceinline__ void ld4g(float* f, gcf_t p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = *reinterpret_cast<const __attribute__((address_space(1))) f4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}

// ---- row-wise helpers: thread tid handles row tid / TPR, columns [CPT * (tid % TPR), +CPT) -----------------------------
// sums over the TPR consecutive lanes that share a row as DPP butterflies inside a 16-lane row
template <int CTRL>
__device__ __forceinline__ float se_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int se_dpp(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
#define SE_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define SE_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define SE_DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i  inside each group of 8
#define SE_DPP_MIRROR 0x140       // lane i <-> 15 - i inside each row of 16
template <int TPR>
__device__ __forceinline__ float row_sum(float v) {
    v += se_dpp<SE_DPP_XOR1>(v);
    v += se_dpp<SE_DPP_XOR2>(v);
    if (TPR >= 8) v += se_dpp<SE_DPP_HALF_MIRROR>(v);
    if (TPR >= 16) v += se_dpp<SE_DPP_MIRROR>(v);
    return v;
}
template <int TPR>
__device__ __forceinline__ float row_max(float v) {
    v = fmaxf(v, se_dpp<SE_DPP_XOR1>(v));
    v = fmaxf(v, se_dpp<SE_DPP_XOR2>(v));
    if (TPR >= 8) v = fmaxf(v, se_dpp<SE_DPP_HALF_MIRROR>(v));
    if (TPR >= 16) v = fmaxf(v, se_dpp<SE_DPP_MIRROR>(v));
    return v;
}
template <int TPR>
__device__ __forceinline__ int row_sum_i(int v) {
    v += se_dpp<SE_DPP_XOR1>(v);
    v += se_dpp<SE_DPP_XOR2>(v);
    if (TPR >= 8) v += se_dpp<SE_DPP_HALF_MIRROR>(v);
    if (TPR >= 16) v += se_dpp<SE_DPP_MIRROR>(v);
    return v;
}

__device__ __forceinline__ void ld4(float* f, const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
}

// ---- item decode ------------------------------------------------------------------------------------------------------
// kind 0: tiles of short sequences (power-of-two slots)   1: one whole long sequence (nt tiles)
// kind 2 / 3: the FIRST two tiles / the remaining tiles of a long sequence SPLIT over two workgroups (re_sasrec_batch_prep with
// split_long; only when every item of the plan gets a workgroup of its own, so both halves are resident at once).  The second half
// attends to the first half's keys: per block, the first half publishes its k, v (forward) and the second half its partial
// dK, dV for the first half's rows (backward) through the tape, with a flag per block.  Only TWO parts ever exist (S <= 64).
struct EncItem {
    int tile0, nt, kind;
};
__device__ __forceinline__ EncItem enc_item(const EncPlan& P, int wi) {
    const int w = P.items[wi];
    return EncItem{w & 0xFFFFFF, (w >> 24) & 0xF, (w >> 28) & 0xF};
}
// the work items of one workgroup: item k of workgroup b is b + k * grid on even rounds, (k + 1) * grid - 1 - b on odd ones
// (items are sorted by size, largest first: the snake order pairs a workgroup's large item with a small one)
__device__ __forceinline__ int enc_item_of(int k, int bid, int grid) { return (k & 1) ? (k + 1) * grid - 1 - bid : k * grid + bid; }

__device__ __forceinline__ int enc_item_npre(const EncItem& it) { return it.kind == 3 ? 2 : 0; }   // prefix key tiles of a second half

// ---- hand-over between the two workgroups of a split sequence.  They may sit on different XCDs, whose L2s are not coherent for
// ordinary accesses inside a kernel: the handed-over tiles and the flags go through device-scope (sc1) accesses, which are.
// Producer: tile stores -> workgroup barrier with vmcnt(0) -> one flag store.  Consumer: one thread polls the flag (bounded: a
// producer that never comes -- it cannot, both are resident -- would otherwise hang the GPU; the error word is set instead),
// barrier, tile loads.  The consumer clears the flag after use: the tape is reusable by the next launch.
__device__ __forceinline__ void enc_flag_set(float* tape_flags, int64_t tile, int word) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(tape_flags) + tile * EP_FLAG_WORDS + word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void enc_flag_wait(float* tape_flags, int64_t tile, int word, int64_t err_word) {
    unsigned* f = reinterpret_cast<unsigned*>(tape_flags) + tile * EP_FLAG_WORDS + word;
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1 << 21)) {   // ~1 s
            __hip_atomic_store(reinterpret_cast<unsigned*>(tape_flags) + err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
    __hip_atomic_store(f, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A plan with split sequences is only valid for a launch in which EVERY item has a workgroup of its own: the halves wait for each
// other, so both must be resident (with grid < n_items the snake order can hand one workgroup both halves -- a certain time-out).
// The plan's workgroup count and the step's grid are independent ABI arguments; this is the check that they agree.  A mismatch
// sets the tape's error word (re: sasrec_tape_errors / check_handover) and, where the launch has a loss word, makes it NaN; the
// launch then does nothing (stale gradients are applied by a captured step's optimizer, but the loss and the error word say so).
__device__ __forceinline__ bool enc_split_plan_rejected(const EncPlan& PL, float* tape_flags, int64_t err_word, float* loss) {
    if (PL.hdr[5] <= 0 || PL.hdr[0] <= (int)gridDim.x) return false;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (tape_flags) __hip_atomic_store(reinterpret_cast<unsigned*>(tape_flags) + err_word, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (loss) loss[0] = __builtin_nanf("");
    }
    return true;
}

template <int D>
__device__ __forceinline__ void tile_store_coh(const float* tile, float* g, int nrows, int tid) {
    tid = enc_opaque(tid);
    using C = EC<D>;
    for (int f = tid; f < nrows * D; f += C::NT)
        __hip_atomic_store(g + f, tile[(f / D) * C::LS + (f % D)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int D>
__device__ __forceinline__ void tile_load_coh(float* tile, const float* g, int nrows, int tid) {
    tid = enc_opaque(tid);
    using C = EC<D>;
    for (int f = tid; f < nrows * D; f += C::NT)
        tile[(f / D) * C::LS + (f % D)] = __hip_atomic_load(g + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int D>
__device__ __forceinline__ void tile_add_coh(float* tile, const float* g, int nrows, int tid) {
    tid = enc_opaque(tid);
    using C = EC<D>;
    for (int f = tid; f < nrows * D; f += C::NT)
        tile[(f / D) * C::LS + (f % D)] += __hip_atomic_load(g + f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Forward -> backward inside ONE launch (enc_step_k): what the backward's prologue would otherwise re-read from memory a moment after
// the forward produced it -- the item's row metadata, lastLN's statistics and the upstream gradient rows (all LDS).  nullptr: separate launches.
struct EncHandoff {
    int *gid, *first, *pad, *sid, *start;
    float *mean, *rstd;     // [ROWS]
    float* du;              // [ROWS][LS] tile: d loss / d u of the item's rows
};

// fills s_gid (b * S + s or -1), s_first (virtual pad keys of the row's sequence), s_pad (1 = pad or dummy row) for the item's rows
template <int D>
__device__ __forceinline__ void enc_decode(const EncPlan& P, const EncItem& it, const int64_t* __restrict__ seq, int tid, int* s_gid,
                                           int* s_first, int* s_pad) {
    if (tid < EC<D>::ROWS) {
        int gid = -1, first = 0;
        if (tid < 16 * it.nt) {
            const int2 rm = P.rowmap[it.tile0 * 16 + tid];
            gid = rm.x; first = rm.y;
        }
        s_gid[tid] = gid;
        s_first[tid] = first;
        s_pad[tid] = (gid < 0) ? 1 : (seq[gid] == 0);
    }
}
// key tiles a row tile attends to: its own tile (short sequences share an item, never a tile's attention) or every tile up to
// its own (one long sequence, causal)
__device__ __forceinline__ int enc_kt_lo(const EncItem& it, int tt) { return it.kind ? 0 : tt; }

// ---- GEMM pieces ----------------------------------------------------------------------------------------------------------
// C[row][16 strip + c] = sum_k A[row][k] B[k][16 strip + c], k over D, for the wave's row tiles tt = t * WR + wr < nt.
// A: LDS tile [ROWS][LS] (k contiguous).  bf[4 q + i] = B[k = 16 q + 4 g + i][n = 16 strip + c] supplied by the caller.
// epi(row, value) is called for the lane's column.  Two accumulators per tile (even / odd steps): a lone tile's MFMAs would
// otherwise wait 40 cycles on each other instead of issuing every 32.
template <int D, int NTW, class Epi>
__device__ __forceinline__ void gemm_rows_n(const float* A, const float (&bf)[D / 4], int lane, int wr, Epi epi) {
    using C = EC<D>;
    const int g = lane >> 4, c = lane & 15;
    f32x4 acc[NTW][2];
    float af[NTW][D / 4];
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tt = t * C::WR + wr;
        acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < D / 16; ++q) ld4(&af[t][4 * q], A + (16 * tt + c) * C::LS + 16 * q + 4 * g);
    }
    __builtin_amdgcn_sched_barrier(0);   // all fragment loads first, then the MFMAs back to back
#pragma unroll
    for (int s = 0; s < D / 4; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t][s & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[t][s], bf[s], acc[t][s & 1], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NTW; ++t) {
        const int tt = t * C::WR + wr;
        { const int rb_ = enc_opaque(16 * tt + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) epi(rb_ + j, acc[t][0][j] + acc[t][1][j]); }
    }
    __builtin_amdgcn_sched_barrier(0);
}
// (wr and nt are wave-uniform scalars: the wave's active tiles are t = 0 .. ntw-1, chosen by a scalar branch -- no per-MFMA predicates)
template <int D, class Epi>
__device__ __forceinline__ void gemm_rows(const float* A, const float (&bf)[D / 4], int lane, int wr, int nt, Epi epi) {
    using C = EC<D>;
    static_assert(C::RT == 2, "two row tiles per wave");
    const int ntw = (nt - wr + C::WR - 1) / C::WR;
    if (ntw >= 2) gemm_rows_n<D, 2>(A, bf, lane, wr, epi);
    else if (ntw == 1) gemm_rows_n<D, 1>(A, bf, lane, wr, epi);
}

// weight fragment for y = x W^T: B[k][n] = W[n][k], W row-major [D][D] in global memory (k contiguous: 16-byte loads)
template <int D>
__device__ __forceinline__ void wfrag_t(float (&bf)[D / 4], const float* W, int strip, int lane) {
    // (the laundered pointer makes these ordinary loads: as loads from a `const __restrict__` kernel argument they are invariant, and
    // the compiler re-issues them right in front of their first use -- a full L2 round trip inside every product -- instead of
    // keeping 16 registers live across the phases in between)
    gcf_t p = g_launder(W) + (16 * strip + (lane & 15)) * D + 4 * (lane >> 4);
#pragma unroll
    for (int q = 0; q < D / 16; ++q) ld4g(&bf[4 * q], p + 16 * q);
}
// weight fragment for dx = dy W: B[k][n] = W[k][n] (k strided)
template <int D>
__device__ __forceinline__ void wfrag_n(float (&bf)[D / 4], const float* W, int strip, int lane) {
    gcf_t p = g_launder(W) + (4 * (lane >> 4)) * D + 16 * strip + (lane & 15);
#pragma unroll
    for (int q = 0; q < D / 16; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) bf[4 * q + i] = p[(16 * q + i) * D];
}

// Score-type product over the item's (row tile, key tile) pairs: T[i][j] = sum_d A[i][d] B[j][d] for i in tile tt, j in tile kt,
// both operands LDS tiles [.][LS] with d contiguous.  Pairs are dealt round-robin over the 8 waves.  epi(row, key column, value).
// A CHAINED item (the later rows of a sequence longer than the LDS holds, enc_fwd.hip) has `npre` prefix key tiles in Bpre: key
// columns [0, 16 npre) are the prefix rows, its own keys follow.
template <int D, class Epi>
__device__ __forceinline__ void gemm_pairs(const float* A, const float* B, int lane, int wave, const EncItem& it, Epi epi,
                                           const float* Bpre = nullptr, int npre = 0) {
    using C = EC<D>;
    const int g = lane >> 4, c = lane & 15;
    int p = 0;
    for (int tt = 0; tt < it.nt; ++tt)
        for (int kk = (npre ? 0 : enc_kt_lo(it, tt)); kk <= npre + tt; ++kk, ++p) {
            if ((p & (C::NW - 1)) != wave) continue;
            const float* Bt = kk < npre ? Bpre + 16 * kk * C::LS : B + 16 * (kk - npre) * C::LS;
            float af[D / 4], bf[D / 4];
#pragma unroll
            for (int q = 0; q < D / 16; ++q) {
                ld4(&af[4 * q], A + (16 * tt + c) * C::LS + 16 * q + 4 * g);
                ld4(&bf[4 * q], Bt + c * C::LS + 16 * q + 4 * g);
            }
            f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < D / 4; s += 2) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s + 1], bf[s + 1], a1, 0, 0, 0);
            }
            { const int rb_ = enc_opaque(16 * tt + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j) epi(rb_ + j, 16 * kk + c, a0[j] + a1[j]); }
            __builtin_amdgcn_sched_barrier(0);
        }
}

// O[i][16 strip + c] = sum_j T[i][j] X[j][16 strip + c] over the key tiles of row tile tt (T: [ROWS][PLS] probabilities-type tile,
// X: [ROWS][LS]) for the wave's row tiles; with prefix tiles (Xpre, npre) the key columns [0, 16 npre) multiply Xpre.  epi(row, value).
template <int D, class Epi>
__device__ __forceinline__ void gemm_tx(const float* T, const float* X, int lane, int wr, int strip, const EncItem& it, Epi epi,
                                        const float* Xpre = nullptr, int npre = 0) {
    using C = EC<D>;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < C::RT; ++t) {
        const int tt = t * C::WR + wr;
        if (tt >= it.nt) continue;
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
        for (int q = (npre ? 0 : enc_kt_lo(it, tt)); q <= npre + tt; ++q) {
            const float* Xt = q < npre ? Xpre + 16 * q * C::LS : X + 16 * (q - npre) * C::LS;
            float af[4], bf[4];
            ld4(af, T + (16 * tt + c) * C::PLS + 16 * q + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) bf[i] = Xt[(4 * g + i) * C::LS + 16 * strip + c];
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], a1, 0, 0, 0);
        }
        { const int rb_ = enc_opaque(16 * tt + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) epi(rb_ + j, a0[j] + a1[j]); }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// O[j][16 strip + c] = sum_i T[i][j] X[i][16 strip + c] for key tile kt = the wave's tiles, i over the row tiles that attend to kt
// (kt itself, or kt .. nt-1 for a long sequence); the item's own keys sit at key columns 16 npre + ...  epi(row j, value).
template <int D, class Epi>
__device__ __forceinline__ void gemm_ttx(const float* T, const float* X, int lane, int wr, int strip, const EncItem& it, Epi epi, int npre = 0) {
    using C = EC<D>;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < C::RT; ++t) {
        const int kt = t * C::WR + wr;
        if (kt >= it.nt) continue;
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
        const int qhi = it.kind ? it.nt - 1 : kt;
        for (int q = kt; q <= qhi; ++q) {
            float af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = T[(16 * q + 4 * g + i) * C::PLS + 16 * (npre + kt) + c];
                bf[i] = X[(16 * q + 4 * g + i) * C::LS + 16 * strip + c];
            }
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], a1, 0, 0, 0);
        }
        { const int rb_ = enc_opaque(16 * kt + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) epi(rb_ + j, a0[j] + a1[j]); }
    }
    __builtin_amdgcn_sched_barrier(0);
}
// The same contraction for the PREFIX key tiles of a chained item: O[j][.] for j in prefix tile pt = the wave's tiles < npre,
// i over ALL the item's row tiles (every row of the item comes after the prefix).  epi(prefix row j, value).
template <int D, class Epi>
__device__ __forceinline__ void gemm_ttx_pre(const float* T, const float* X, int lane, int wr, int strip, const EncItem& it, int npre, Epi epi) {
    using C = EC<D>;
    const int g = lane >> 4, c = lane & 15;
#pragma unroll
    for (int t = 0; t < C::RT; ++t) {
        const int pt = t * C::WR + wr;
        if (pt >= npre) continue;
        f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
        for (int q = 0; q < it.nt; ++q) {
            float af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = T[(16 * q + 4 * g + i) * C::PLS + 16 * pt + c];
                bf[i] = X[(16 * q + 4 * g + i) * C::LS + 16 * strip + c];
            }
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[0], bf[0], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[1], bf[1], a1, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[2], bf[2], a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[3], bf[3], a1, 0, 0, 0);
        }
        { const int rb_ = enc_opaque(16 * pt + 4 * g);
#pragma unroll
        for (int j = 0; j < 4; ++j) epi(rb_ + j, a0[j] + a1[j]); }
    }
    __builtin_amdgcn_sched_barrier(0);
}

// ---- tile <-> global ------------------------------------------------------------------------------------------------------
// An item's rows are contiguous in the compact (tape) arrays: [16 nt][D] starting at row 16 tile0.
template <int D>
struct TileRegs {
    f32x4 v[EC<D>::ROWS * (D / 4) / EC<D>::NT];   // (native vectors: arrays of HIP's float4 struct tend to end up in scratch memory)
};
template <int D>
__device__ __forceinline__ void tile_fetch(TileRegs<D>& R, const float* g, int nrows, int tid) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
    gcf_t gp = g_launder(g);   // (pins the request where it is written)
#pragma unroll
    for (int q = 0; q < C::ROWS * (D / 4) / C::NT; ++q) {
        const int f = q * C::NT + tid;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (f < nrows * (D / 4)) ld4g(t, gp + 4 * f);
        R.v[q] = (f32x4){t[0], t[1], t[2], t[3]};
    }
}
template <int D>
__device__ __forceinline__ void tile_commit(float* tile, const TileRegs<D>& R, int nrows, int tid) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
#pragma unroll
    for (int q = 0; q < C::ROWS * (D / 4) / C::NT; ++q) {
        const int f = q * C::NT + tid;
        if (f < nrows * (D / 4)) *reinterpret_cast<f32x4*>(tile + (f / (D / 4)) * C::LS + 4 * (f % (D / 4))) = R.v[q];
    }
}
template <int D>
__device__ __forceinline__ void tile_store(const float* tile, float* __restrict__ g, int nrows, int tid) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
    for (int f = tid; f < nrows * (D / 4); f += C::NT)
        reinterpret_cast<float4*>(g)[f] = *reinterpret_cast<const float4*>(tile + (f / (D / 4)) * C::LS + 4 * (f % (D / 4)));
}
template <int D>
__device__ __forceinline__ void tile_store(const float* tile, float* __restrict__ g, int nrows, int tid, float scale) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
    for (int f = tid; f < nrows * (D / 4); f += C::NT) {
        const float4 v = *reinterpret_cast<const float4*>(tile + (f / (D / 4)) * C::LS + 4 * (f % (D / 4)));
        reinterpret_cast<float4*>(g)[f] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
}
// tile[r][:] += g[r][:] for nrows contiguous rows (the partial sums a chained item's later rows left for its earlier rows)
template <int D>
__device__ __forceinline__ void tile_add_global(float* tile, const float* g, int nrows, int tid) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
    gcf_t gp = g_launder(g);
    for (int f = tid; f < nrows * (D / 4); f += C::NT) {
        float t[4];
        ld4g(t, gp + 4 * f);
        float* d = tile + (f / (D / 4)) * C::LS + 4 * (f % (D / 4));
        d[0] += t[0]; d[1] += t[1]; d[2] += t[2]; d[3] += t[3];
    }
}
// rows of a [B*S][D] matrix selected by s_gid (dummy rows read as zero / are not written)
template <int D>
__device__ __forceinline__ void tile_fetch_gid(TileRegs<D>& R, const float* g, const int* s_gid, int nrows, int tid) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
    gcf_t gp = g_launder(g);
#pragma unroll
    for (int q = 0; q < C::ROWS * (D / 4) / C::NT; ++q) {
        const int f = q * C::NT + tid;
        float t[4] = {0.f, 0.f, 0.f, 0.f};
        if (f < nrows * (D / 4)) {
            const int gid = s_gid[f / (D / 4)];
            if (gid >= 0) ld4g(t, gp + (int64_t)gid * D + 4 * (f % (D / 4)));
        }
        R.v[q] = (f32x4){t[0], t[1], t[2], t[3]};
    }
}
// rows of a table by per-row index: the loss head's E[pos], E[neg] (index 0 = the table's padding row; unconditional, clamped loads:
// a predicated load is waited for where it is issued)
template <int D>
__device__ __forceinline__ void tile_fetch_rows(TileRegs<D>& R, const float* table, const int* s_row, int nrows, int tid) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
    gcf_t gp = g_launder(table);
    const int last = nrows * (D / 4) - 1;
#pragma unroll
    for (int q = 0; q < C::ROWS * (D / 4) / C::NT; ++q) {
        int f = q * C::NT + tid;
        f = f < last ? f : last;
        float t[4];
        ld4g(t, gp + (int64_t)s_row[f / (D / 4)] * D + 4 * (f % (D / 4)));
        R.v[q] = (f32x4){t[0], t[1], t[2], t[3]};
    }
}
template <int D>
__device__ __forceinline__ void tile_store_gid(const float* tile, float* __restrict__ g, const int* s_gid, int nrows, int tid, float scale = 1.0f) {
    tid = enc_opaque(tid);   // (per-thread addresses recomputed at the call, not carried across the block loop: enc_opaque)
    using C = EC<D>;
    for (int f = tid; f < nrows * (D / 4); f += C::NT) {
        const int r = f / (D / 4), c4 = f % (D / 4);
        const int gid = s_gid[r];
        if (gid >= 0) {
            const float4 v = *reinterpret_cast<const float4*>(tile + r * C::LS + 4 * c4);
            reinterpret_cast<float4*>(g + (int64_t)gid * D)[c4] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
        }
    }
}

// LayerNorm of this thread's row slice (eps 1e-8, biased variance -- nn.LayerNorm, SASRec/main.py:89,94,106)
template <int D>
__device__ __forceinline__ void ln_row(const float* src, float* dst, const float* __restrict__ gw, const float* __restrict__ gb, int tid,
                                       float& mean, float& rstd) {
    using C = EC<D>;
    const int r = tid / C::TPR, c0 = (tid % C::TPR) * C::CPT;
    float x[C::CPT];
#pragma unroll
    for (int q = 0; q < C::CPT / 4; ++q) ld4(&x[4 * q], src + r * C::LS + c0 + 4 * q);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < C::CPT; ++i) s += x[i];
    mean = row_sum<C::TPR>(s) * (1.0f / D);
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < C::CPT; ++i) { const float d = x[i] - mean; q2 = fmaf(d, d, q2); }
    rstd = 1.0f / sqrtf(row_sum<C::TPR>(q2) * (1.0f / D) + 1e-8f);
#pragma unroll
    for (int i = 0; i < C::CPT; ++i) dst[r * C::LS + c0 + i] = (x[i] - mean) * rstd * gw[c0 + i] + gb[c0 + i];
}

// ---- a block's small parameters (LayerNorm scale / shift, the six biases: 10 vectors of D floats) are staged in LDS one block
// ahead: read where they are used straight from global memory, every one of them is an exposed L2 round trip on the item's
// critical path -- and the counter that waits for it (vmcnt, in order) also waits for every older weight-fragment request.
//   slot: 0 ln_a_w 1 ln_a_b 2 bq 3 bk 4 bv 5 bo 6 ln_f_w 7 ln_f_b 8 b1 9 b2
#define EP_NPAR 10
template <int D>
struct ParRegs {
    float v[(EP_NPAR * D + EC<D>::NT - 1) / EC<D>::NT];
};
template <int D>
__device__ __forceinline__ void par_fetch(ParRegs<D>& R, const SasrecBlockParams& W, int tid) {
    tid = enc_opaque(tid);
    using C = EC<D>;
#pragma unroll
    for (int q = 0; q < (EP_NPAR * D + C::NT - 1) / C::NT; ++q) {
        const int e = q * C::NT + tid;
        const int v = e / D, cc = e % D;
        const float* p = (v == 0) ? W.ln_a_w : (v == 1) ? W.ln_a_b : (v < 5) ? W.in_b + (v - 2) * D : (v == 5) ? W.out_b
                       : (v == 6) ? W.ln_f_w : (v == 7) ? W.ln_f_b : (v == 8) ? W.b1 : W.b2;
        asm volatile("" : "+v"(p));   // (an opaque address: an ordinary load, issued here)
        R.v[q] = (e < EP_NPAR * D) ? ((gcf_t)p)[cc] : 0.f;
    }
}
template <int D>
__device__ __forceinline__ void par_commit(float* dst, const ParRegs<D>& R, int tid) {
    tid = enc_opaque(tid);
    using C = EC<D>;
#pragma unroll
    for (int q = 0; q < (EP_NPAR * D + C::NT - 1) / C::NT; ++q) {
        const int e = q * C::NT + tid;
        if (e < EP_NPAR * D) dst[e] = R.v[q];
    }
}

static bool se_fill_params(SasrecParams& P, const float* const* bp, int64_t L, const float* last_w, const float* last_b) {
    if (!bp || !last_w || !last_b || L < 1 || L > SE_MAX_BLOCKS) return false;
    for (int64_t l = 0; l < L; ++l) {
        const float* const* q = bp + 12 * l;
        for (int i = 0; i < 12; ++i)
            if (!q[i]) return false;
        P.blk[l] = SasrecBlockParams{q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8], q[9], q[10], q[11]};
    }
    P.last_w = last_w;
    P.last_b = last_b;
    return true;
}
