// K6/K7 training step of the SASRec encoder at D = 64: forward + criterion + backward of a 16-row tile by FOUR WAVES, one per
// 16-feature strip, with the tile's activations in registers (4 per lane and matrix) and every linear map on the XDL pipe.
//
// Reference restated: SASRec/main.py:178-193 (encode), :163-176 (after_one_block), :31-50 (PointWiseFeedForward), :199-215 (fit).
//
// Why this shape.  A Beauty-shaped batch of 512 is ~300 tiles: less than one wave of work per SIMD, so a launch lasts as long as ONE
// tile's dependent chain.  The workgroup-per-item kernel (enc_step.hip) spent 118 k cycles per item in ~75 barrier-separated phases
// over LDS tiles; one wave per tile with everything in registers (enc_wave.hip, the step before this one) has no barriers but is
// ONE in-order stream of ~20 k instructions at 4 - 5 cycles each.  Here a tile's stream is cut four ways by FEATURE STRIP: wave s
// owns features [16 s, 16 s + 16) of every activation, so all element-wise work (LayerNorm, dropout hashes, residuals, masks,
// operand splits, column sums) is a quarter per wave, each wave issues 6 MFMAs per linear map instead of 24 and holds 16 weight
// registers instead of 64, and what crosses waves is small: the bf16 planes of a map's input (4 KB per tile through LDS), per-token
// partial sums (LayerNorm, the criterion's dots) and the partial score tiles of the attention products.
//
// LAYOUTS (lane = 16 g + c, wave = 4 tile + strip s).  Of a tile matrix M[token][feature] a wave holds, in 4 registers,
//   T strip: m[j] = M[c][16 s + 4 g + j]      (the lane's token is c)         F strip: m[j] = M[4 g + j][16 s + c]   (the lane's feature)
// and a 16 x 16 score-type matrix S[query][key] as Rt: s[j] = S[c][4 g + j] or R: s[j] = S[4 g + j][c].
// v_mfma_f32_16x16x32_bf16 puts the OUTER index of both operands on lane & 15 and 8 k values on lane >> 4; with k slot i of step q
// standing for feature 16 (2 q + (i >> 2)) + 4 g + (i & 3), the four T strips of a matrix, lane by lane, ARE its two k-step
// fragments: the B operand of Y^T = W X^T (result: T strip s of Y from the weight rows of strip s as A) and the A operand of
// V = X Wv^T (result: F strip of V, which O^T = V^T P^T wants as A).  Weights are pre-arranged once per step in that k order, both
// orientations (enc_tile_prep_k).  Products over tokens or over the 16 features of a strip use v_mfma_f32_16x16x16_bf16: P V, dS K,
// dS^T Q, P^T dO strip-locally; q k^T and dO v^T as partial score tiles per strip, summed over the four waves in strip order.
// Arithmetic: operands are split hi = bf16(x), mid = bf16(x - hi); a product is hi.hi + hi.mid + mid.hi with fp32 accumulation
// (<= 3.01 * 2^-18 of sum |a b| per product: inside the 1e-4 parity bound); all row-wise arithmetic is fp32.
//
// WORK.  Workgroup = 4 waves = ONE tile of the plan (blockIdx = compact tile number; what a tile is -- shared by short sequences,
// or tile t of a sequence of nt tiles -- is read off the plan's row map), so the ~300 tiles of a batch spread over every CU.
// Tiles of short sequences are independent.  A long sequence's tile t attends to the keys of tiles 0..t, which other WORKGROUPS
// compute: per block, a tile publishes its k, v rows (forward) and its partial dK, dV for earlier tiles (backward) with device-scope
// stores, drains them (s_waitcnt vmcnt(0)), barriers, and one lane stores the launch's EPOCH into the tile's flag word; a consumer
// polls that word (device-scope load, bounded), barriers, and reads the rows with device-scope loads (MI355X guide: "handoff-flag").
// A forward wait is for a LOWER block index and a backward wait for tiles of the same sequence; the plan lays the long sequences'
// tiles first, so all of them are resident from the start as long as there are fewer of them than resident workgroups -- the plan
// kernel checks exactly that (hdr[7] = 1) and otherwise leaves the step to the workgroup-per-item kernel (enc_step.hip), which is
// launched right behind this one and returns at once when this one has run.  Sums over tiles are taken in tile order: deterministic.
// The weight gradients stay in enc_wgrad.hip (tape X, A, O, Y, HR + the six dY arrays); bias / LayerNorm gradients are column
// sums over a strip's tokens, written per TILE to the slab.
#include <math.h>

#include "enc_fwd_item.h"
#include "enc_tile_prep.h"

// ---- operand splits ---------------------------------------------------------------------------------------------------------
struct Op64 { tl_u32x4 h[2], m[2]; };   // K = 64 features: 2 steps x 8 bf16, hi and mid planes
struct Op16 { tl_u32x2 h, m; };         // K = 16: 4 bf16
__device__ __forceinline__ void tl_split4(const f32x4& x, Op16& o) {
    unsigned h0, m0, h1, m1;
    tl_split2(x[0], x[1], h0, m0);
    tl_split2(x[2], x[3], h1, m1);
    o.h = (tl_u32x2){h0, h1}; o.m = (tl_u32x2){m0, m1};
}
__device__ __forceinline__ f32x4 tl_mfma32(tl_u32x4 a, tl_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(tl_bf16x8, a), __builtin_bit_cast(tl_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 tl_mfma16(tl_u32x2 a, tl_u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(tl_s16x4, a), __builtin_bit_cast(tl_s16x4, b), c, 0, 0, 0);
}
// C[m][n] = sum over 64 features of A[m][.] B[.][n]: the small cross terms first, then hi.hi
__device__ __forceinline__ f32x4 tl_mm64(const Op64& a, const Op64& b) {
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        acc = tl_mfma32(a.m[q], b.h[q], acc);
        acc = tl_mfma32(a.h[q], b.m[q], acc);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) acc = tl_mfma32(a.h[q], b.h[q], acc);
    return acc;
}
__device__ __forceinline__ f32x4 tl_mm16(const Op16& a, const Op16& b, f32x4 acc) {
    acc = tl_mfma16(a.m, b.h, acc);
    acc = tl_mfma16(a.h, b.m, acc);
    return tl_mfma16(a.h, b.h, acc);
}
// the wave's strip of a weight matrix's fragments (enc_tile_prep_k): 4 x 16 bytes per lane
__device__ __forceinline__ void tl_wload(Op64& w, const uint32_t* __restrict__ wf, int l, int mat, int orient, int s, int lane) {
    const tl_u32x4* p = reinterpret_cast<const tl_u32x4*>(wf + ((size_t)(l * 6 + mat) * 2 + orient) * TL_FRAG_WORDS) + (size_t)s * 4 * 64 + lane;
    w.h[0] = p[0]; w.m[0] = p[64]; w.h[1] = p[128]; w.m[1] = p[192];
}

// ---- operand slots: a wave writes the bf16 planes of its strip, every wave of the tile reads all four ------------------------
__device__ __forceinline__ void tl_put(float* ob, int slot, int lane, int s, const f32x4& x) {
    Op16 o;
    tl_split4(x, o);
    tl_u32x2* p = reinterpret_cast<tl_u32x2*>(ob) + ((slot * 2) * 64 + lane) * 4 + s;
    p[0] = o.h;
    p[64 * 4] = o.m;
}
__device__ __forceinline__ void tl_get(const float* ob, int slot, int lane, Op64& o) {
    const tl_u32x4* p = reinterpret_cast<const tl_u32x4*>(reinterpret_cast<const tl_u32x2*>(ob) + ((slot * 2) * 64 + lane) * 4);
    o.h[0] = p[0]; o.h[1] = p[1];
    o.m[0] = p[128]; o.m[1] = p[129];
}
// LDS-only workgroup barrier (the wave's own LDS operations are complete; vector-memory operations stay in flight)
__device__ __forceinline__ void tl_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- sums ---------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float tl_gsum(float v) {   // over the four lanes that share a token (lane ^ 16, lane ^ 32)
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float tl_gmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
// Column sum of a T strip over its 16 tokens: two halving steps (each lane keeps half of its registers and adds the partner's copy of
// that half), two plain steps.  Every lane ends with the sum of register j(c) = 2 b3 + b2 (bits of c), i.e. of feature 16 s + 4 g + j.
#define TL_DPP_ROR8 0x128
__device__ __forceinline__ float tl_colsum(const f32x4& x, int c) {
    const bool b3 = (c & 8) != 0, b2 = (c & 4) != 0;
    const float y0 = (b3 ? x[2] : x[0]) + se_dpp<TL_DPP_ROR8>(b3 ? x[0] : x[2]);            // partner c ^ 8
    const float y1 = (b3 ? x[3] : x[1]) + se_dpp<TL_DPP_ROR8>(b3 ? x[1] : x[3]);
    float z = (b2 ? y1 : y0) + se_dpp<SE_DPP_HALF_MIRROR>(b2 ? y0 : y1);                    // partner c ^ 7
    z += se_dpp<SE_DPP_XOR1>(z);
    z += se_dpp<SE_DPP_XOR2>(z);
    return z;
}
// 16 x 16 transpose Rt -> R through the wave's LDS scratch: in[j] = M[c][4 g + j] -> out[j] = M[4 g + j][c]
__device__ __forceinline__ void tl_tr16(float* scr, int c, int g, const f32x4& in, f32x4& out) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (earlier readers of the scratch are done: one wave, in order)
    *reinterpret_cast<f32x4*>(scr + c * 20 + 4 * g) = in;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = scr[(4 * g + j) * 20 + c];
}
__device__ __forceinline__ f32x4 tl_ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ void tl_st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
// device-scope (sc1) accesses for what crosses workgroups: the XCDs' L2s are not coherent for ordinary accesses inside a kernel
__device__ __forceinline__ float tl_ldc(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void tl_stc(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ f32x4 tl_ld4c(const float* p) { return (f32x4){tl_ldc(p), tl_ldc(p + 1), tl_ldc(p + 2), tl_ldc(p + 3)}; }
__device__ __forceinline__ void tl_st4c(float* p, const f32x4& v) { tl_stc(p, v[0]); tl_stc(p + 1, v[1]); tl_stc(p + 2, v[2]); tl_stc(p + 3, v[3]); }
// flag word of (tile, slot): slots 0 .. 3 = forward k, v of block l published, 4 .. 7 = backward partial dK, dV of block l published
__device__ __forceinline__ void tl_flag_set(float* flags, int64_t tile, int word, unsigned epoch) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(flags) + tile * EP_FLAG_WORDS + word, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tl_flag_wait(float* flags, int64_t tile, int word, unsigned epoch, int64_t err_word) {
    unsigned* f = reinterpret_cast<unsigned*>(flags) + tile * EP_FLAG_WORDS + word;
    int spins = 0;
    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 21)) {   // ~1 s: the producer never came (it cannot: see the residency rule above) -- say so instead of hanging the GPU
            __hip_atomic_store(reinterpret_cast<unsigned*>(flags) + err_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
    }
}

__global__ __launch_bounds__(256) void enc_tile_prep_k(SasrecParams P, int L, uint32_t* __restrict__ wf, unsigned* __restrict__ epoch) {
    tl_prep_thread(P, L, wf, epoch, blockIdx.x * 256 + threadIdx.x);
}

#ifdef TL_PROFILE
#define TL_MARKS 96
#define TL_MARK_BLOCKS 8
__device__ unsigned long long g_tile_marks[TL_MARK_BLOCKS * TL_MARKS];   // the stamps of workgroups 0 .. 7 (the tiles of the first long sequences)
extern "C" int re_dbg_enc_marks_wave(unsigned long long* out) {          // (workgroup 0's)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_marks), sizeof(unsigned long long) * TL_MARKS) == hipSuccess ? 0 : 1;
}
extern "C" int re_dbg_enc_marks_blocks(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_marks), sizeof(unsigned long long) * TL_MARK_BLOCKS * TL_MARKS) == hipSuccess ? 0 : 1;
}
#define TL_MARK() do { if (blockIdx.x < TL_MARK_BLOCKS && tid == 0 && k == 0 && mk < TL_MARKS) g_tile_marks[blockIdx.x * TL_MARKS + mk] = __builtin_amdgcn_s_memtime(); ++mk; } while (0)
#else
#define TL_MARK() do { } while (0)
#endif

struct TlArgs {
    SeEmbed em;
    const int64_t* seq;
    int B, S, L;
    float drop_scale;
    uint32_t thresh, seed;
    float *u, *tape;
    EncTape T;
    const void* planp;
    EncHead H;
    float *dOut, *gtape, *slab;
    float emb_scale;
    const uint32_t* wf;
    float* xch;
};

// The work of one tile.  MULTI = a tile of a sequence longer than 16 rows (key tiles 0 .. tt, hand-overs between workgroups); the other
// instantiation -- a tile shared by short sequences: ONE key tile, nothing crosses workgroups -- is what most tiles of a batch run.
template <bool MULTI>
__device__ __forceinline__ void tl_tile(const TlArgs& A, float* lds, const int tile, const int n_tiles, const int2 rm0) {
    const SeEmbed& em = A.em;
    const int64_t* __restrict__ seq = A.seq;
    const int B = A.B, S = A.S, L = A.L;
    const float drop_scale = A.drop_scale, emb_scale = A.emb_scale;
    const uint32_t thresh = A.thresh, seed = A.seed;
    float* __restrict__ u = A.u;
    float* __restrict__ tape = A.tape;
    const EncTape& T = A.T;
    const EncHead& H = A.H;
    float* __restrict__ dOut = A.dOut;
    float* __restrict__ gtape = A.gtape;
    float* __restrict__ slab = A.slab;
    const uint32_t* __restrict__ wf = A.wf;
    float* __restrict__ xch = A.xch;
    const EncPlan PL = enc_plan_view(A.planp, B, S);
    constexpr int NK = MULTI ? 4 : 1;                    // key tiles a token can attend to
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4, s = wave;
    float* s_par = lds;
    float* ob = s_par + (TL_NPAR * L + 2) * TL_D;
    float* red = ob + TL_OB;
    float* sm = red + TL_RED;
    float* scr = sm + TL_SM + s * 320;
    float* flags = tape + T.off_FLAGS;
    const int64_t ferr = enc_plan_max_tiles(B, S) * EP_FLAG_WORDS;
    const unsigned epoch = MULTI ? reinterpret_cast<const unsigned*>(flags)[ferr + 1] : 0u;
    const float inv_sqrt_d = 0.125f;
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    const int tofs = c * TL_D + 16 * s + 4 * g;          // T strip: float4 at row c
    const int fofs = 4 * g * TL_D + 16 * s + c;          // F strip: element j at + j * D
    {
        const int k = 0; (void)k;
        // ---- what this tile is: row 0 of a tile is always a real row; a sequence of more than 16 rows owns whole tiles, in order
        const int64_t row0 = (int64_t)tile * 16;                     // compact row of the tile's first row
        constexpr bool multi = MULTI;                                // tile tt of a sequence of nt tiles: k, v and dK, dV cross workgroups
        const int span0 = rm0.x >= 0 ? S - rm0.y : 1;
        const int tt = multi ? (rm0.x % S - rm0.y) / 16 : 0, nt = multi ? (span0 + 15) / 16 : 1;
        const int64_t irow0 = row0 - 16 * tt;                        // compact row of the sequence's first row
        const int64_t tile_s0 = tile - tt;                           // the sequence's first tile
        const int rb = (int)row0 * TL_D, irb = (int)irow0 * TL_D, nrd = (int)(NR * TL_D);   // 32-bit offsets from a block's tape / gradient-tape base
        const int nkx = multi ? tt + 1 : 1;                          // score-exchange rounds (key tiles 0 .. tt of a long sequence, else the own tile)
        constexpr bool live = true;
        int mk = 0; (void)mk;
        int slot = 0, rp = 0;                                        // operand slot / partial-buffer rotation (uniform over the workgroup)
        TL_MARK();
        // ---- the lane's token
        const int2 rm = PL.rowmap[row0 + c];
        const int gid = rm.x, n_out = rm.y;
        const int io = 16 * tt + c;                                  // item-local row
        int st = 0;
        int64_t item = 0;
        if (gid >= 0) {
            const int sid = gid / S;
            st = io - (gid - sid * S - n_out);
            item = seq[gid];
        }
        const bool real = item > 0 && item < em.R;
        const bool dead = gid < 0 || item == 0;                      // pad or dummy row: x' = 0 after every block
        const unsigned span = gid >= 0 ? (unsigned)(io - st) : 0u;
        int64_t hpr = 0, hng = 0;                                    // the loss head's indices: requested now, used after the last block
        if (gid >= 0) { hpr = H.pos[gid] + H.e_off; hng = H.neg[gid] + H.e_off; }
        const float hgs = 1.0f / (float)H.count[0];
        // Weight strips: two register sets (16 registers each), each requested one product ahead of its use
        Op64 wa, wb;
        tl_wload(wa, wf, 0, 0, 0, s, lane);                          // block 0's Wq strip
        // the loss head's table rows: requested now (their indices are known), used after the last block
        const bool hreal = item > 0 && item < H.R;
        const bool hok = hreal && hpr > 0 && hpr < H.R && hng > 0 && hng < H.R;
        if (!hok) { hpr = 0; hng = 0; }
        const f32x4 hep = tl_ld4(H.E + hpr * TL_D + 16 * s + 4 * g), hen = tl_ld4(H.E + hng * TL_D + 16 * s + 4 * g);
        // ---- x0 = E[item] sqrt(D) + P[position], dropout (SASRec/main.py:181-187): the rows are requested, the parameters staged meanwhile
        f32x4 x = (f32x4){0.f, 0.f, 0.f, 0.f}, xe = x, xp = x;
        unsigned emask = 0xFu;
        if (real) {
            xe = tl_ld4(em.E + item * TL_D + 16 * s + 4 * g);
            xp = tl_ld4(em.P + (int64_t)(gid % S) * TL_D + 16 * s + 4 * g);
        }
        // the small parameters of every block and lastLN into LDS (behind the requests above; enc_tile_prep_k gathered them into one block)
        {
            const float* pb = reinterpret_cast<const float*>(wf + (size_t)L * 6 * 2 * TL_FRAG_WORDS);
            for (int e = tid; e < (TL_NPAR * L + 2) * TL_D; e += 256) s_par[e] = pb[e];
        }
        __syncthreads();
        if (real) {
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = fmaf(xe[j], em.scale, xp[j]);
            if (thresh) {
                emask = 0u;
                const uint32_t e0 = (uint32_t)((int64_t)gid * TL_D + 16 * s + 4 * g);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool kp = re_keep(seed, RE_STREAM_EMBED, e0 + j, thresh);
                    emask |= (kp ? 1u : 0u) << j;
                    x[j] = kp ? x[j] * drop_scale : 0.f;
                }
            }
        }
        const uint32_t el0 = (uint32_t)((int64_t)gid * TL_D + 16 * s + 4 * g);   // element index of x[0] in a [B, S, D] tensor

// LayerNorm statistics of the lane's token from the four strips (Chan's merge of per-strip mean / M2: the two-pass accuracy without a
// second exchange).  One barrier.
#define TL_LN_STATS(X, MEAN, RSTD)                                                                                        \
        do {                                                                                                              \
            const float ms_ = tl_gsum((X[0] + X[1]) + (X[2] + X[3])) * (1.0f / 16);                                       \
            float q2_ = 0.f;                                                                                              \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) { const float d_ = X[j] - ms_; q2_ = fmaf(d_, d_, q2_); }       \
            q2_ = tl_gsum(q2_);                                                                                           \
            float* smb_ = sm + rp * (4 * 16 * 4);                                                                         \
            if (g == 0) *reinterpret_cast<float2*>(smb_ + (s * 16 + c) * 4) = make_float2(ms_, q2_);                      \
            tl_sync();                                                                                                    \
            const float2 p0_ = *reinterpret_cast<const float2*>(smb_ + (0 * 16 + c) * 4);                                 \
            const float2 p1_ = *reinterpret_cast<const float2*>(smb_ + (1 * 16 + c) * 4);                                 \
            const float2 p2_ = *reinterpret_cast<const float2*>(smb_ + (2 * 16 + c) * 4);                                 \
            const float2 p3_ = *reinterpret_cast<const float2*>(smb_ + (3 * 16 + c) * 4);                                 \
            rp ^= 1;                                                                                                      \
            MEAN = ((p0_.x + p1_.x) + (p2_.x + p3_.x)) * 0.25f;                                                           \
            const float a0_ = p0_.x - MEAN, a1_ = p1_.x - MEAN, a2_ = p2_.x - MEAN, a3_ = p3_.x - MEAN;                    \
            const float m2_ = ((p0_.y + p1_.y) + (p2_.y + p3_.y)) + 16.0f * ((a0_ * a0_ + a1_ * a1_) + (a2_ * a2_ + a3_ * a3_)); \
            RSTD = 1.0f / sqrtf(m2_ * (1.0f / TL_D) + 1e-8f);                                                             \
        } while (0)
// two per-token scalars summed over the four strips (in strip order).  One barrier.
#define TL_PAIR_SUM(A, Bv)                                                                                                \
        do {                                                                                                              \
            float* smb_ = sm + rp * (4 * 16 * 4);                                                                         \
            if (g == 0) *reinterpret_cast<float2*>(smb_ + (s * 16 + c) * 4) = make_float2(A, Bv);                         \
            tl_sync();                                                                                                    \
            const float2 p0_ = *reinterpret_cast<const float2*>(smb_ + (0 * 16 + c) * 4);                                 \
            const float2 p1_ = *reinterpret_cast<const float2*>(smb_ + (1 * 16 + c) * 4);                                 \
            const float2 p2_ = *reinterpret_cast<const float2*>(smb_ + (2 * 16 + c) * 4);                                 \
            const float2 p3_ = *reinterpret_cast<const float2*>(smb_ + (3 * 16 + c) * 4);                                 \
            rp ^= 1;                                                                                                      \
            A = (p0_.x + p1_.x) + (p2_.x + p3_.x);                                                                        \
            Bv = (p0_.y + p1_.y) + (p2_.y + p3_.y);                                                                       \
        } while (0)
// the partial 16 x 16 score tiles of NK key tiles (+ two per-token scalars) summed over the four strips, in strip order.  One barrier.
#define TL_TILES_SUM(PT, NK, A, Bv)                                                                                       \
        do {                                                                                                              \
            float* rb_ = red + rp * (4 * 64 * 16);                                                                        \
            float* smb_ = sm + rp * (4 * 16 * 4);                                                                         \
            _Pragma("unroll") for (int kt_ = 0; kt_ < NK; ++kt_)                                                           \
                if (kt_ < (NK)) tl_st4(rb_ + ((s * 64 + lane) * 4 + kt_) * 4, PT[kt_]);                                   \
            if (g == 0) *reinterpret_cast<float2*>(smb_ + (s * 16 + c) * 4) = make_float2(A, Bv);                         \
            tl_sync();                                                                                                    \
            _Pragma("unroll") for (int kt_ = 0; kt_ < NK; ++kt_)                                                           \
                if (kt_ < (NK)) {                                                                                         \
                    const f32x4 t0_ = tl_ld4(rb_ + ((0 * 64 + lane) * 4 + kt_) * 4), t1_ = tl_ld4(rb_ + ((1 * 64 + lane) * 4 + kt_) * 4); \
                    const f32x4 t2_ = tl_ld4(rb_ + ((2 * 64 + lane) * 4 + kt_) * 4), t3_ = tl_ld4(rb_ + ((3 * 64 + lane) * 4 + kt_) * 4); \
                    _Pragma("unroll") for (int j = 0; j < 4; ++j) PT[kt_][j] = (t0_[j] + t1_[j]) + (t2_[j] + t3_[j]);     \
                }                                                                                                         \
            const float2 p0_ = *reinterpret_cast<const float2*>(smb_ + (0 * 16 + c) * 4);                                 \
            const float2 p1_ = *reinterpret_cast<const float2*>(smb_ + (1 * 16 + c) * 4);                                 \
            const float2 p2_ = *reinterpret_cast<const float2*>(smb_ + (2 * 16 + c) * 4);                                 \
            const float2 p3_ = *reinterpret_cast<const float2*>(smb_ + (3 * 16 + c) * 4);                                 \
            rp ^= 1;                                                                                                      \
            A = (p0_.x + p1_.x) + (p2_.x + p3_.x);                                                                        \
            Bv = (p0_.y + p1_.y) + (p2_.y + p3_.y);                                                                       \
        } while (0)
#define TL_NEXT_SLOT() (slot = (slot == 2) ? 0 : slot + 1)

        // =================================================== forward ===================================================
        for (int l = 0; l < L; ++l) {
            float* tp = tape + (int64_t)l * T.per_block;
            const float* par = s_par + l * TL_NPAR * TL_D + 16 * s + 4 * g;     // the lane's four features of a parameter vector
            TL_MARK();
            // ---- a = LN_a(x)
            f32x4 a;
            {
                float mean, rstd;
                TL_LN_STATS(x, mean, rstd);
                const f32x4 gw = tl_ld4(par + 0 * TL_D), gb = tl_ld4(par + 1 * TL_D);
#pragma unroll
                for (int j = 0; j < 4; ++j) a[j] = fmaf((x[j] - mean) * rstd, gw[j], gb[j]);
                if (live && g == 0) *reinterpret_cast<float2*>(tp + ((int)T.off_SA + 2 * ((int)row0 + c))) = make_float2(mean, rstd);   // (written four times over: identical values)
            }
            if (l == 0) tl_wload(wb, wf, 0, 1, 0, s, lane);   // Wk (later blocks: requested at the end of the block before)
            // ---- q = a Wq^T + bq, k = x Wk^T + bk (T strips), v = x Wv^T + bv (F strip)
            const int sa = slot; TL_NEXT_SLOT();
            const int sx = slot; TL_NEXT_SLOT();
            tl_put(ob, sa, lane, s, a);
            tl_put(ob, sx, lane, s, x);
            if (live) { tl_st4(tp + ((int)T.off_X + rb + tofs), x); tl_st4(tp + ((int)T.off_A + rb + tofs), a); }
            tl_sync();
            TL_MARK();
            Op64 ao, xo;
            tl_get(ob, sa, lane, ao);
            tl_get(ob, sx, lane, xo);
            f32x4 q = tl_mm64(wa, ao);
            tl_wload(wa, wf, l, 2, 0, s, lane);           // Wv
            const f32x4 bq = tl_ld4(par + 2 * TL_D), bk = tl_ld4(par + 3 * TL_D), bvt = tl_ld4(par + 4 * TL_D);
            f32x4 kk = tl_mm64(wb, xo);
            tl_wload(wb, wf, l, 3, 0, s, lane);           // Wo
            f32x4 vf = tl_mm64(xo, wa);
            tl_wload(wa, wf, l, 4, 0, s, lane);           // W1
            const float bvs = s_par[l * TL_NPAR * TL_D + 4 * TL_D + 16 * s + c];
#pragma unroll
            for (int j = 0; j < 4; ++j) { q[j] += bq[j]; kk[j] += bk[j]; vf[j] += bvs; }
            tl_st4(tp + ((int)T.off_Q + rb + tofs), q);
            if (multi && tt + 1 < nt) {                   // later tiles of the sequence read these rows: device-scope stores, then the flag
                tl_st4c(tp + ((int)T.off_K + rb + tofs), kk);
#pragma unroll
                for (int j = 0; j < 4; ++j) tl_stc(tp + ((int)T.off_V + rb + fofs + j * TL_D), vf[j]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) tl_flag_set(flags, tile, l, epoch);
            } else {
                tl_st4(tp + ((int)T.off_K + rb + tofs), kk);
#pragma unroll
                for (int j = 0; j < 4; ++j) tp[(int)T.off_V + rb + fofs + j * TL_D] = vf[j];
            }
            if (tt > 0) {                                 // the earlier tiles' k, v of this block
                if (tid == 0)
                    for (int kt = 0; kt < tt; ++kt) tl_flag_wait(flags, tile_s0 + kt, l, epoch, ferr);
                __syncthreads();
            }
            TL_MARK();
            // ---- scores Rt(S)[kt] = q k^T / sqrt(D): partial over the strip's 16 features, summed over the strips; with them the
            //      pad key's score q.b_k and the count of kept pad keys (each of the n_out pad keys has its own dropout bit)
            Op16 qo;
            tl_split4(q, qo);
            float dqb = tl_gsum((q[0] * bk[0] + q[1] * bk[1]) + (q[2] * bk[2] + q[3] * bk[3]));
            float cntf = 0.f;
            if (thresh) {
                int cnt = 0;
                const uint32_t e0 = (uint32_t)((int64_t)gid * S);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int jj = 16 * i + 4 * s + g;
                    cnt += (jj < n_out && gid >= 0 && re_keep(seed, RE_STREAM_ATTN(l), e0 + (uint32_t)jj, thresh)) ? 1 : 0;
                }
                cntf = tl_gsum((float)cnt);
            }
            f32x4 p[NK];
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                p[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kt >= nkx) continue;                   // (workgroup-uniform: every wave takes part in every round of the item)
                const int ktc = multi ? kt : tt;
                Op16 ko;
                if (ktc == tt) tl_split4(kk, ko);
                else tl_split4(tl_ld4c(tp + ((int)T.off_K + irb + 16 * TL_D * ktc + tofs)), ko);
                p[kt] = tl_mm16(ko, qo, p[kt]);
            }
            TL_TILES_SUM(p, nkx, dqb, cntf);
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                if (kt >= nkx) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) p[kt][j] *= inv_sqrt_d;
            }
            // (kind 0: the one round above was for key tile tt; keep it at index tt - klo = 0)
            // ---- softmax over the keys of the token's own sequence up to itself, plus the virtual pad key (multiplicity n_out,
            //      score q.b_k / sqrt(D), value b_v), dropout on the probabilities -- enc_fwd_item.h, same arithmetic; every strip
            //      computes it (the probabilities are needed by all four)
            float mx = -INFINITY;
            unsigned okm = 0u;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                if (kt >= nkx) continue;                   // (uniform: a tile of short sequences has one key tile)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ktc = multi ? kt : tt;
                    const int jo = 16 * ktc + 4 * g + j;
                    const bool ok = gid >= 0 && (unsigned)(jo - st) <= span;
                    okm |= (ok ? 1u : 0u) << (4 * kt + j);
                    mx = fmaxf(mx, ok ? p[kt][j] : -INFINITY);
                }
            }
            const float spad = (gid >= 0 && n_out > 0) ? dqb * inv_sqrt_d : -INFINITY;
            mx = tl_gmax(fmaxf(mx, spad));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                if (kt >= nkx) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p[kt][j] = ((okm >> (4 * kt + j)) & 1u) ? expf(p[kt][j] - mx) : 0.f;
                    sum += p[kt][j];
                }
            }
            sum = tl_gsum(sum);
            const float epad = (spad == -INFINITY) ? 0.f : expf(spad - mx);
            sum += (float)n_out * epad;
            const float inv = (gid >= 0) ? 1.0f / sum : 0.f;
            const float ppad = epad * inv;
            const float kept = thresh ? cntf * drop_scale : (float)n_out;
            const float wvv = (gid >= 0) ? ppad * kept : 0.f;
            if (live && s == 0 && g == 0) *reinterpret_cast<float2*>(tp + ((int)T.off_PP + 2 * ((int)row0 + c))) = make_float2(ppad, wvv);
            unsigned amask = 0xFFFFu;
            f32x4 pd[NK];
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                pd[kt] = p[kt];
                if (kt >= nkx) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) p[kt][j] *= inv;
                pd[kt] = p[kt];
                const int ktc = multi ? kt : tt;
                if (s == 0)   // pre-dropout probabilities (0 outside the token's window): row c, key columns 16 kt + 4 g ..
                    tl_st4(tp + ((int)T.off_P + ((int)row0 + c) * EP_PW + 16 * ktc + 4 * g), p[kt]);
            }
            if (thresh) {
                amask = 0u;
                const uint32_t e0 = (uint32_t)((int64_t)gid * S + n_out - st);   // + item-local key row: the key's position in the sequence
#pragma unroll
                for (int kt = 0; kt < NK; ++kt) {
                    const int ktc = multi ? kt : tt;
                    if (kt >= nkx || ktc > tt) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool kp = re_keep(seed, RE_STREAM_ATTN(l), e0 + (uint32_t)(16 * ktc + 4 * g + j), thresh);
                        amask |= (kp ? 1u : 0u) << (4 * kt + j);
                        pd[kt][j] = kp ? p[kt][j] * drop_scale : 0.f;
                    }
                }
            }
            TL_MARK();
            // ---- o = Pd v + w b_v   (T strip of o = sum over key tiles of F(v strip)-as-A x Rt(Pd)-as-B)
            f32x4 o = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                const int ktc = multi ? kt : tt;
                if (kt >= nkx || ktc > tt) continue;
                Op16 po, vo;
                tl_split4(pd[kt], po);
                f32x4 vt = vf;
                if (ktc != tt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) vt[j] = tl_ldc(tp + ((int)T.off_V + irb + 16 * TL_D * ktc + fofs + j * TL_D));
                }
                tl_split4(vt, vo);
                o = tl_mm16(vo, po, o);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = fmaf(wvv, bvt[j], o[j]);
            // ---- x1 = o Wo^T + bo + x
            const int so = slot; TL_NEXT_SLOT();
            tl_put(ob, so, lane, s, o);
            if (live) tl_st4(tp + ((int)T.off_O + rb + tofs), o);
            tl_sync();
            TL_MARK();
            f32x4 x1;
            {
                Op64 oo;
                tl_get(ob, so, lane, oo);
                x1 = tl_mm64(wb, oo);
                tl_wload(wb, wf, l, 5, 0, s, lane);       // W2
                const f32x4 bo = tl_ld4(par + 5 * TL_D);
#pragma unroll
                for (int j = 0; j < 4; ++j) x1[j] += bo[j] + x[j];
            }
            if (live) tl_st4(tp + ((int)T.off_X1 + rb + tofs), x1);
            // ---- y = LN_f(x1)
            f32x4 y;
            {
                float mean, rstd;
                TL_LN_STATS(x1, mean, rstd);
                const f32x4 gw = tl_ld4(par + 6 * TL_D), gb = tl_ld4(par + 7 * TL_D);
#pragma unroll
                for (int j = 0; j < 4; ++j) y[j] = fmaf((x1[j] - mean) * rstd, gw[j], gb[j]);
                if (live && g == 0) *reinterpret_cast<float2*>(tp + ((int)T.off_SF + 2 * ((int)row0 + c))) = make_float2(mean, rstd);
            }
            // ---- hr = relu(dropout1(y W1^T + b1))
            const int sy = slot; TL_NEXT_SLOT();
            tl_put(ob, sy, lane, s, y);
            if (live) tl_st4(tp + ((int)T.off_Y + rb + tofs), y);
            tl_sync();
            TL_MARK();
            f32x4 hr;
            unsigned hmask = 0u;
            {
                Op64 yo;
                tl_get(ob, sy, lane, yo);
                hr = tl_mm64(wa, yo);
                if (l + 1 < L) tl_wload(wa, wf, l + 1, 0, 0, s, lane);   // the next block's Wq,
                else tl_wload(wa, wf, L - 1, 5, 1, s, lane);            // or the backward's first: W2 of the last block
                const f32x4 b1 = tl_ld4(par + 8 * TL_D);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = hr[j] + b1[j];
                    if (thresh) v = re_keep(seed, RE_STREAM_FFN1(l), el0 + j, thresh) ? v * drop_scale : 0.f;
                    hr[j] = fmaxf(v, 0.f);
                    hmask |= (hr[j] > 0.f ? 1u : 0u) << j;
                }
            }
            // ---- x' = dropout2(hr W2^T + b2) + y, pad rows zeroed
            const int sh = slot; TL_NEXT_SLOT();
            tl_put(ob, sh, lane, s, hr);
            if (live) tl_st4(tp + ((int)T.off_HR + rb + tofs), hr);
            tl_sync();
            TL_MARK();
            unsigned m2 = 0xFu;
            {
                Op64 ho;
                tl_get(ob, sh, lane, ho);
                const f32x4 z = tl_mm64(wb, ho);
                if (l + 1 < L) tl_wload(wb, wf, l + 1, 1, 0, s, lane);   // the next block's Wk, or the backward's second: W1
                else tl_wload(wb, wf, L - 1, 4, 1, s, lane);
                const f32x4 b2 = tl_ld4(par + 9 * TL_D);
                if (thresh) m2 = 0u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = z[j] + b2[j];
                    if (thresh) {
                        const bool kp = re_keep(seed, RE_STREAM_FFN2(l), el0 + j, thresh);
                        m2 |= (kp ? 1u : 0u) << j;
                        v = kp ? v * drop_scale : 0.f;
                    }
                    x[j] = dead ? 0.f : v + y[j];
                }
            }
            // the block's mask bits for the backward: one word per lane and strip ([tile][4 strips][64] words of the tape's mask array)
            if (live) reinterpret_cast<uint32_t*>(tp)[(int)T.off_MK + (int)row0 * 16 + s * 64 + lane] = m2 | (hmask << 4) | (amask << 8);
        }
        TL_MARK();
        // ---- u = LN_last(x_L)
        float rstd_l;
        f32x4 uu, xh, glw;
        {
            float mean_l;
            TL_LN_STATS(x, mean_l, rstd_l);
            const float* pl = s_par + TL_NPAR * L * TL_D + 16 * s + 4 * g;
            glw = tl_ld4(pl);
            const f32x4 bb = tl_ld4(pl + TL_D);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xh[j] = (x[j] - mean_l) * rstd_l;
                uu[j] = fmaf(xh[j], glw[j], bb[j]);
            }
        }
        if (live && gid >= 0) tl_st4(u + (int64_t)gid * TL_D + 16 * s + 4 * g, uu);
        // ---- loss head (SASRec/main.py:199-215): pl = <u, E[pos]>, nl = <u, E[neg]>; the rows' gradient contributions and keys
        f32x4 du;
        float head_loss = 0.f;
        {
            const int64_t pr = hpr, ng = hng;
            const bool realh = hreal, ok = hok;
            const f32x4 ep = hep, en = hen;
            float pl = tl_gsum((uu[0] * ep[0] + uu[1] * ep[1]) + (uu[2] * ep[2] + uu[3] * ep[3]));
            float nl = tl_gsum((uu[0] * en[0] + uu[1] * en[1]) + (uu[2] * en[2] + uu[3] * en[3]));
            TL_PAIR_SUM(pl, nl);
            float dpl, dnl;
            if (H.kind == RE_LOSS_BCE) { dpl = -re_sigmoid(-pl) * hgs; dnl = re_sigmoid(nl) * hgs; }
            else { const float sg = re_sigmoid(nl - pl) * hgs; dpl = -sg; dnl = sg; }
            if (!ok) { dpl = 0.f; dnl = 0.f; }
            f32x4 gp, gn;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                du[j] = fmaf(dpl, ep[j], dnl * en[j]);
                gp[j] = dpl * uu[j];
                gn[j] = dnl * uu[j];
            }
            if (live) {
                tl_st4(H.dU_rows + row0 * TL_D + tofs, du);
                if (ok) {
                    tl_st4(H.g_rows + (NR + row0) * TL_D + tofs, gp);
                    tl_st4(H.g_rows + (2 * NR + row0) * TL_D + tofs, gn);
                }
                if (s == 0 && g == 0) {
                    H.keys[row0 + c] = realh ? (int)item : 0;
                    H.keys[NR + row0 + c] = (int)pr;
                    H.keys[2 * NR + row0 + c] = (int)ng;
                    if (ok) head_loss = (H.kind == RE_LOSS_BCE) ? re_softplus(-pl) + re_softplus(nl) : re_softplus(nl - pl);
                }
            }
        }
        // =================================================== backward ===================================================
        __syncthreads();   // (with vmcnt(0): what the forward left on the tape for OTHER waves of the workgroup -- P, the pad-key weights, the
                           //  LayerNorm statistics, a long sequence's k / v rows -- is in L2 before the backward reads it)
        TL_MARK();
        float* srow = slab + (size_t)tile * L * EG_NVEC * TL_D + 16 * s + 4 * g + ((c >> 3) * 2 + ((c >> 2) & 1));   // the lane's column-sum feature
        const bool cs_w = live && (c & 3) == 0;              // one lane of every four holds a column sum to write
#define TL_COLSUM(V, VAL) do { const float cs_ = tl_colsum(VAL, c); if (cs_w) srow[(size_t)(l * EG_NVEC + (V)) * TL_D] = cs_; } while (0)
        f32x4 dx;
        {
            // lastLN: dgamma, dbeta, dx_L = rstd (d - mean(d) - xh mean(d xh)), d = du gamma
            const int l = L - 1;
            f32x4 t;
#pragma unroll
            for (int j = 0; j < 4; ++j) t[j] = du[j] * xh[j];
            TL_COLSUM(10, t);
            TL_COLSUM(11, du);
            f32x4 d;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { d[j] = du[j] * glw[j]; s1 += d[j]; s2 = fmaf(d[j], xh[j], s2); }
            s1 = tl_gsum(s1); s2 = tl_gsum(s2);
            TL_PAIR_SUM(s1, s2);
            s1 *= 1.0f / TL_D; s2 *= 1.0f / TL_D;
#pragma unroll
            for (int j = 0; j < 4; ++j) dx[j] = rstd_l * (d[j] - s1 - xh[j] * s2);
        }
        // a block's mask word, x1 and per-token statistics are requested one block ahead (the first block's here, behind the barrier)
        unsigned mw_n = reinterpret_cast<const uint32_t*>(tape + (int64_t)(L - 1) * T.per_block + T.off_MK)[row0 * 16 + s * 64 + lane];
        f32x4 x1_n = tl_ld4(tape + (int64_t)(L - 1) * T.per_block + T.off_X1 + row0 * TL_D + tofs);
        float2 sf_n = *reinterpret_cast<const float2*>(tape + (int64_t)(L - 1) * T.per_block + T.off_SF + (row0 + c) * 2);
        float2 ppw_n = *reinterpret_cast<const float2*>(tape + (int64_t)(L - 1) * T.per_block + T.off_PP + (row0 + c) * 2);
        for (int l = L - 1; l >= 0; --l) {
            const float* tp = tape + (int64_t)l * T.per_block;
            const float* par = s_par + l * TL_NPAR * TL_D + 16 * s + 4 * g;
            float* gp = gtape + (int64_t)l * EG_NMAT * NR * TL_D + row0 * TL_D + tofs;
            TL_MARK();
            if (l != L - 1 && cs_w) { srow[(size_t)(l * EG_NVEC + 10) * TL_D] = 0.f; srow[(size_t)(l * EG_NVEC + 11) * TL_D] = 0.f; }
            const unsigned mw = mw_n, amask = mw >> 8;
            const f32x4 x1 = x1_n;
            const float2 sf = sf_n, ppw = ppw_n;
            if (l > 0) {
                const float* tn = tp - T.per_block;
                mw_n = reinterpret_cast<const uint32_t*>(tn)[(int)T.off_MK + (int)row0 * 16 + s * 64 + lane];
                x1_n = tl_ld4(tn + ((int)T.off_X1 + rb + tofs));
                sf_n = *reinterpret_cast<const float2*>(tn + ((int)T.off_SF + 2 * ((int)row0 + c)));
                ppw_n = *reinterpret_cast<const float2*>(tn + ((int)T.off_PP + 2 * ((int)row0 + c)));
            }
            // ---- pad mask of the block output, dO2 = dX' * dropout2 mask
            f32x4 dz;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dx[j] = dead ? 0.f : dx[j];
                dz[j] = !thresh ? dx[j] : ((mw >> j) & 1u) ? dx[j] * drop_scale : 0.f;
            }
            const int s0 = slot; TL_NEXT_SLOT();
            tl_put(ob, s0, lane, s, dz);
            if (live) tl_st4(gp + 0 * nrd, dz);
            TL_COLSUM(5, dz);
            tl_sync();
            // ---- A. dH = (dO2 W2) * (hr > 0) * scale
            f32x4 dh;
            {
                Op64 oo;
                tl_get(ob, s0, lane, oo);
                dh = tl_mm64(wa, oo);
                tl_wload(wa, wf, l, 3, 1, s, lane);       // Wo
#pragma unroll
                for (int j = 0; j < 4; ++j) dh[j] = ((mw >> (4 + j)) & 1u) ? dh[j] * drop_scale : 0.f;
            }
            const int s1s = slot; TL_NEXT_SLOT();
            tl_put(ob, s1s, lane, s, dh);
            if (live) tl_st4(gp + 1 * nrd, dh);
            TL_COLSUM(4, dh);
            tl_sync();
            TL_MARK();
            // ---- B. dY = dH W1 + dX'
            f32x4 dy;
            {
                Op64 oo;
                tl_get(ob, s1s, lane, oo);
                dy = tl_mm64(wb, oo);
                tl_wload(wb, wf, l, 0, 1, s, lane);       // Wq
#pragma unroll
                for (int j = 0; j < 4; ++j) dy[j] += dx[j];
            }
            // ---- C. LN_f backward: dgamma_f, dbeta_f, dX1
            f32x4 dx1;
            {
                f32x4 xh1, t, d;
                float s1 = 0.f, s2 = 0.f;
                const f32x4 gw = tl_ld4(par + 6 * TL_D);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xh1[j] = (x1[j] - sf.x) * sf.y;
                    t[j] = dy[j] * xh1[j];
                    d[j] = dy[j] * gw[j];
                    s1 += d[j];
                    s2 = fmaf(d[j], xh1[j], s2);
                }
                TL_COLSUM(8, t);
                TL_COLSUM(9, dy);
                s1 = tl_gsum(s1); s2 = tl_gsum(s2);
                TL_PAIR_SUM(s1, s2);
                s1 *= 1.0f / TL_D; s2 *= 1.0f / TL_D;
#pragma unroll
                for (int j = 0; j < 4; ++j) dx1[j] = sf.y * (d[j] - s1 - xh1[j] * s2);
            }
            const int s2s = slot; TL_NEXT_SLOT();
            tl_put(ob, s2s, lane, s, dx1);
            if (live) tl_st4(gp + 2 * nrd, dx1);
            TL_COLSUM(3, dx1);
            // q and the own tile's k in F layout, v in T layout, x and LN_a's statistics: requested here, used behind the products below
            f32x4 qf, kf_own;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qf[j] = tp[(int)T.off_Q + rb + fofs + j * TL_D];
                kf_own[j] = tp[(int)T.off_K + rb + fofs + j * TL_D];
            }
            const f32x4 vt_own = tl_ld4(tp + ((int)T.off_V + rb + tofs));
            const f32x4 qt = tl_ld4(tp + ((int)T.off_Q + rb + tofs));
            const f32x4 xx = tl_ld4(tp + ((int)T.off_X + rb + tofs));
            const float2 sa_ = *reinterpret_cast<const float2*>(tp + ((int)T.off_SA + 2 * ((int)row0 + c)));
            tl_sync();
            TL_MARK();
            // ---- D. dO = dX1 Wo, as T strip and as F strip (the same fragments, operands swapped)
            f32x4 dO, dOf;
            {
                Op64 oo;
                tl_get(ob, s2s, lane, oo);
                dO = tl_mm64(wa, oo);
                dOf = tl_mm64(oo, wa);
                tl_wload(wa, wf, l, 1, 1, s, lane);       // Wk
            }
            // ---- E. attention backward (enc_bwd_item.h, same arithmetic): dP = dO v^T as partial tiles per strip
            const float ppad = ppw.x, wvv = ppw.y;
            const f32x4 bvt = tl_ld4(par + 4 * TL_D), bk = tl_ld4(par + 3 * TL_D);
            float tdot = tl_gsum((dO[0] * bvt[0] + dO[1] * bvt[1]) + (dO[2] * bvt[2] + dO[3] * bvt[3]));
            float acc_bv;
            {
                f32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = wvv * dO[j];
                acc_bv = tl_colsum(t, c);                  // d b_v through the virtual pad key: sum_i w_i dO_i
            }
            Op16 doo;
            tl_split4(dO, doo);
            f32x4 p[NK], ds[NK];                            // ds: first dP, then dS (the dropped probabilities are rebuilt from the mask bits where they are used)
            float srw = 0.f;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                p[kt] = ds[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kt >= nkx) continue;
                const int ktc = multi ? kt : tt;
                p[kt] = tl_ld4(tp + ((int)T.off_P + ((int)row0 + c) * EP_PW + 16 * ktc + 4 * g));
                Op16 vo;
                tl_split4(ktc == tt ? vt_own : tl_ld4c(tp + ((int)T.off_V + irb + 16 * TL_D * ktc + tofs)), vo);
                ds[kt] = tl_mm16(vo, doo, ds[kt]);         // partial (dO_i . v_j) for token i = c, keys 4 g + j
            }
            {
                float zb = 0.f;
                TL_TILES_SUM(ds, nkx, tdot, zb);
            }
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                if (kt >= nkx) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mf = !thresh ? 1.0f : ((amask >> (4 * kt + j)) & 1u) ? drop_scale : 0.f;
                    ds[kt][j] = (p[kt][j] != 0.f) ? ds[kt][j] * mf : 0.f;
                    srw = fmaf(ds[kt][j], p[kt][j], srw);
                }
            }
            srw = tl_gsum(srw);
            srw = fmaf(tdot, wvv, srw);                                            // the row dot includes the pad copies
            const float cpad = (wvv * tdot - (float)n_out * ppad * srw) * inv_sqrt_d;   // sum of dS over the pad copies
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                if (kt >= nkx) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) ds[kt][j] = p[kt][j] * (ds[kt][j] - srw) * inv_sqrt_d;
            }
            TL_MARK();
            // dQ = dS K + dS_pad b_k;  per key tile dV_kt = Pd^T dO, dK_kt = dS^T Q  (strip-local: K = tokens)
            Op16 qo4, do4;
            tl_split4(qf, qo4);
            tl_split4(dOf, do4);
            f32x4 dq = (f32x4){0.f, 0.f, 0.f, 0.f}, dk = dq, dv = dq;
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                const int ktc = multi ? kt : tt;
                if (kt >= nkx || ktc > tt) continue;
                f32x4 kf = kf_own;
                if (ktc != tt) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) kf[j] = tl_ldc(tp + ((int)T.off_K + irb + 16 * TL_D * ktc + fofs + j * TL_D));
                }
                Op16 ko, dso, pdr, dsr;
                tl_split4(kf, ko);
                tl_split4(ds[kt], dso);
                f32x4 tr, pdk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pdk[j] = !thresh ? p[kt][j] : ((amask >> (4 * kt + j)) & 1u) ? p[kt][j] * drop_scale : 0.f;
                tl_tr16(scr, c, g, pdk, tr);
                tl_split4(tr, pdr);
                tl_tr16(scr, c, g, ds[kt], tr);
                tl_split4(tr, dsr);
                dq = tl_mm16(ko, dso, dq);                                          // T(dq): features x queries
                const f32x4 pv_ = tl_mm16(do4, pdr, (f32x4){0.f, 0.f, 0.f, 0.f});   // T(dv_kt): features x keys
                const f32x4 pk_ = tl_mm16(qo4, dsr, (f32x4){0.f, 0.f, 0.f, 0.f});   // T(dk_kt)
                if (ktc == tt) { dv = pv_; dk = pk_; }
                else {
                    // this tile's contribution to an EARLIER tile's dV, dK: into that tile's inbox, slot tt - kt - 1 (device-scope stores)
                    float* in = xch + (size_t)(tile_s0 + ktc) * TL_XCH_TILE + (size_t)((tt - ktc - 1) * 2) * 1024 + s * 256 + lane;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { tl_stc(in + j * 64, pv_[j]); tl_stc(in + 1024 + j * 64, pk_[j]); }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) dq[j] = fmaf(cpad, bk[j], dq[j]);
            if (tt > 0) {                                  // published: drained, then the flag
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) tl_flag_set(flags, tile, 4 + l, epoch);
            }
            if (tt + 1 < nt) {                             // the later tiles' partials for these rows, added in tile order
                if (tid == 0)
                    for (int t = tt + 1; t < nt; ++t) tl_flag_wait(flags, tile_s0 + t, 4 + l, epoch, ferr);
                __syncthreads();
                for (int t = tt + 1; t < nt; ++t) {
                    const float* in = xch + (size_t)tile * TL_XCH_TILE + (size_t)((t - tt - 1) * 2) * 1024 + s * 256 + lane;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { dv[j] += tl_ldc(in + j * 64); dk[j] += tl_ldc(in + 1024 + j * 64); }
                }
            }
            // ---- G. dA = dQ Wq;  dX = dX1 + dK Wk + dV Wv + LN_a'(dA)
            const int sq = slot; TL_NEXT_SLOT();
            const int sk = slot; TL_NEXT_SLOT();
            const int sv = slot; TL_NEXT_SLOT();
            tl_put(ob, sq, lane, s, dq);
            tl_put(ob, sk, lane, s, dk);
            tl_put(ob, sv, lane, s, dv);
            if (live) { tl_st4(gp + 3 * nrd, dq); tl_st4(gp + 4 * nrd, dk); tl_st4(gp + 5 * nrd, dv); }
            TL_COLSUM(0, dq);
            {
                f32x4 t;
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = cpad * qt[j];
                const float c1 = tl_colsum(t, c) + tl_colsum(dk, c);   // d b_k: sum_i dS_pad_i q_i + the key rows
                const float c2 = acc_bv + tl_colsum(dv, c);
                if (cs_w) { srow[(size_t)(l * EG_NVEC + 1) * TL_D] = c1; srow[(size_t)(l * EG_NVEC + 2) * TL_D] = c2; }
            }
            tl_sync();
            TL_MARK();
            f32x4 da;
            {
                Op64 oo;
                tl_get(ob, sq, lane, oo);
                da = tl_mm64(wb, oo);
                tl_wload(wb, wf, l, 2, 1, s, lane);       // Wv
                tl_get(ob, sk, lane, oo);
                const f32x4 t1 = tl_mm64(wa, oo);
                if (l > 0) tl_wload(wa, wf, l - 1, 5, 1, s, lane);   // the next block's W2
                tl_get(ob, sv, lane, oo);
                const f32x4 t2 = tl_mm64(wb, oo);
                if (l > 0) tl_wload(wb, wf, l - 1, 4, 1, s, lane);   // ... and W1
#pragma unroll
                for (int j = 0; j < 4; ++j) dx1[j] += t1[j] + t2[j];
            }
            {
                f32x4 xha, t, d;
                float s1 = 0.f, s2 = 0.f;
                const f32x4 gw = tl_ld4(par + 0 * TL_D);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    xha[j] = (xx[j] - sa_.x) * sa_.y;
                    t[j] = da[j] * xha[j];
                    d[j] = da[j] * gw[j];
                    s1 += d[j];
                    s2 = fmaf(d[j], xha[j], s2);
                }
                TL_COLSUM(6, t);
                TL_COLSUM(7, da);
                s1 = tl_gsum(s1); s2 = tl_gsum(s2);
                TL_PAIR_SUM(s1, s2);
                s1 *= 1.0f / TL_D; s2 *= 1.0f / TL_D;
#pragma unroll
                for (int j = 0; j < 4; ++j) dx[j] = dx1[j] + sa_.y * (d[j] - s1 - xha[j] * s2);
            }
        }
        TL_MARK();
        // ---- embedding backward (re_sasrec_embed_bwd fused in): pad rows -> 0, the embedding's dropout mask, * sqrt(D)
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = dead ? 0.f : dx[j];
                if (thresh && !dead) v = ((emask >> j) & 1u) ? v * drop_scale : 0.f;
                dx[j] = v * emb_scale;
            }
            if (gid >= 0) tl_st4(dOut + (int64_t)gid * TL_D + 16 * s + 4 * g, dx);
            tl_st4(H.g_rows + row0 * TL_D + tofs, dx);
        }
        // ---- the tile's loss -> the ticket (strip 0's wave holds it)
        if (wave == 0) {
            head_loss = re_wave_sum(head_loss);
            if (lane == 0) {
                const double part = (double)head_loss;
                const bool finite = part == part && fabs(part) < 4294967296.0;
                const unsigned long long add = finite ? (unsigned long long)(long long)llrint(part * 1073741824.0) : 0ull;
                const unsigned long long old = __hip_atomic_fetch_add(H.acc, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const unsigned long long one = 1ull + (finite ? 0ull : (1ull << 32)) + (old & 0ull);
                const unsigned long long ticket = __hip_atomic_fetch_add(H.acc + 1, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int)(ticket & 0xFFFFFFFFull) == n_tiles - 1) {
                    const unsigned long long tot = __hip_atomic_exchange(H.acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const bool bad = ((ticket + one) >> 32) != 0ull;
                    const int cnt = H.count[0];
                    H.loss[0] = (cnt > 0 && !bad) ? (float)((double)(long long)tot * (1.0 / 1073741824.0) / (double)cnt) : __builtin_nanf("");
                    __hip_atomic_store(H.acc + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256, 2) void enc_tile_step_k(SeEmbed em, const int64_t* __restrict__ seq, int B, int S, int L,
                                                         float drop_scale, uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                         float* __restrict__ tape, EncTape T, const void* __restrict__ planp, EncHead H,
                                                         float* __restrict__ dOut, float* __restrict__ gtape, float* __restrict__ slab,
                                                         const uint32_t* __restrict__ seed_dev, float emb_scale,
                                                         const uint32_t* __restrict__ wf, float* __restrict__ xch) {
    extern __shared__ __align__(16) float lds[];
    const EncPlan PL = enc_plan_view(planp, B, S);
    if (PL.hdr[7] != 1) return;                          // (not a plan for this kernel: the workgroup-per-item kernel behind it runs the step)
    const int n_tiles = PL.hdr[1];
    const int tile = blockIdx.x;
    if (tile >= n_tiles) return;
    if (seed_dev) seed ^= seed_dev[0];
    const TlArgs A{em, seq, B, S, L, drop_scale, thresh, seed, u, tape, T, planp, H, dOut, gtape, slab, emb_scale, wf, xch};
    const int2 rm0 = PL.rowmap[(int64_t)tile * 16];
    if (rm0.x >= 0 && S - rm0.y > 16) tl_tile<true>(A, lds, tile, n_tiles, rm0);
    else tl_tile<false>(A, lds, tile, n_tiles, rm0);
}

int enc_tile_step_launch(const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P, float ds, uint32_t thresh,
                         uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan, int grid, const EncHead& H, float* dx0,
                         float* gtape, float* slab, float scale, uint32_t* wf, float* xch, int prep, hipStream_t s) {
    const EncTape T = enc_tape_layout(B, S, TL_D, L);
    if (prep) {   // (0: the batch preparation launch of this step has written the fragments and advanced the epoch)
        hipLaunchKernelGGL(enc_tile_prep_k, dim3((unsigned)(TL_PREP_THREADS(L) / 256)), dim3(256), 0, s, P, (int)L, wf, enc_tile_epoch(tape, B, S, L));
        if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    }
    const size_t ldsb = tl_lds_floats((int)L) * sizeof(float);
    if (hipFuncSetAttribute((const void*)enc_tile_step_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
    hipLaunchKernelGGL(enc_tile_step_k, dim3(grid), dim3(256), ldsb, s, em, seq, (int)B, (int)S, (int)L, ds, thresh, seed, u, (float*)tape, T, plan, H,
                       dx0, gtape, slab, seed_dev, scale, (const uint32_t*)wf, xch);
    return hipGetLastError() == hipSuccess ? RE_OK : RE_ELAUNCH;
}
