// K6/K7 training step of the SASRec encoder at D = 64, ONE WAVE PER 16-ROW TILE: forward + criterion + backward of a tile run as one
// in-order instruction stream with the tile's activations in REGISTERS; nothing but the keys / values of a long sequence's other
// tiles crosses waves.  (The workgroup-per-item kernel of enc_step.hip spent 86 % of an item's 118 k cycles in ~75 barrier-separated
// phases that eight waves could not overlap -- a Beauty-shaped batch is ~300 tiles on 256 CUs, so a launch lasts ONE item's latency.)
//
// Reference restated: SASRec/main.py:178-193 (encode), :163-176 (after_one_block), :31-50 (PointWiseFeedForward), :199-215 (fit).
//
// LAYOUTS (lane = 16 g + c).  A tile matrix M[token][feature] (16 x 64) lives in 16 registers per lane as
//   T(M): m[4 s + j] = M[c][16 s + 4 g + j]      (the lane's token is c: the C layout of an MFMA whose N index is the token)
//   F(M): m[4 s + j] = M[4 g + j][16 s + c]      (the lane's feature is c: the C layout of an MFMA whose M index is the token)
// and a 16 x 16 score-type matrix S[query][key] as Rt(S): s[j] = S[c][4 g + j] or R(S): s[j] = S[4 g + j][c].
// Both operands of v_mfma_f32_16x16x32_bf16 put their OUTER index (M for A, N for B) on lane & 15 and 8 k values on lane >> 4; the
// k order of a step is free as long as A and B agree.  With k slot i of step q standing for feature 16 (2 q + (i >> 2)) + 4 g + (i & 3),
// registers 8 q .. 8 q + 7 of T(M) ARE the fragment of step q -- as the B operand when the product's N index is the token
// (Y^T = W X^T: the result comes out as T(Y), ready to be the next product's operand) and as the A operand when its M index is
// (V = X Wv^T comes out as F(V), what O^T = V^T P^T wants as A).  So a whole block chains through registers; the weights are
// pre-arranged once per step as fragments in that k order (enc_wave_prep_k: W for y = x W^T, W^T for dx = dy W).
// Products over tokens (K = 16: P V, dS K, dS^T Q, P^T dO) use v_mfma_f32_16x16x16_bf16 on F / R / Rt operands.
// Arithmetic: every operand is split hi = bf16(x), mid = bf16(x - hi); a product is hi.hi + hi.mid + mid.hi on the XDL pipe with fp32
// accumulation (relative error <= 3.01 * 2^-18 of sum |a b|: inside the 1e-4 parity bound); all row-wise
// arithmetic (LayerNorm, softmax, dropout, criterion) is fp32 as in the workgroup kernel.
// Transposes: F(V) -> T(V), T(Q) -> F(Q), T(K) -> F(K) go through the tape (stored in one layout, read in the other: L2 hits);
// Rt -> R of a 16 x 16 tile through 1.25 KB of wave-private LDS.
//
// WORK.  Item = 1..4 tiles (re_sasrec_batch_prep; kinds 0 / 1 only -- no split sequences), workgroup = 4 waves, wave w = tile w.
// Tiles of short sequences (kind 0) are independent.  A long sequence's tile t attends to the keys of tiles 0..t: k, v travel
// through the tape behind one workgroup barrier per block, and the partial dK, dV a later tile has for an earlier one through LDS
// slots, summed by the owner in tile order (deterministic).  The weight gradients stay in enc_wgrad.hip (tape X, A, O, Y, HR and
// the six dY arrays of the gradient tape, as before); bias / LayerNorm gradients are column sums over the lane's tokens.
#include <math.h>

#include "enc_fwd_item.h"

typedef __bf16 wv_bf16x8 __attribute__((ext_vector_type(8)));
typedef short wv_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned wv_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned wv_u32x2 __attribute__((ext_vector_type(2)));

#define WV_NT 256
#define WV_NW 4
#define WV_D 64
#define WV_FRAG_WORDS 4096      // one (block, matrix, orientation): [strip 4][k step 2][plane 2][lane 64] x 16 bytes
#define WV_NPAR 10              // per block: 0 ln_a_w 1 ln_a_b 2 bq 3 bk 4 bv 5 bo 6 ln_f_w 7 ln_f_b 8 b1 9 b2
#define WV_TR 320               // floats of a wave's 16 x 16 transpose scratch (row stride 20)
#define WV_XCH_SLOTS 6          // (t, kt < t) pairs of a 4-tile sequence

size_t enc_wave_wfrag_bytes(int64_t L) { return (size_t)L * 6 * 2 * WV_FRAG_WORDS * 4; }

// ---- operand splits ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned wv_pk(float a, float b) {   // two fp32 -> packed bf16 (round to nearest even), a in the low half
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 v = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, v);
}
struct Op64 { wv_u32x4 h[2], m[2]; };   // K = 64 features: 2 steps x 8 bf16, hi and mid planes
struct Op16 { wv_u32x2 h, m; };         // K = 16 tokens: 4 bf16

__device__ __forceinline__ void wv_split2(float a, float b, unsigned& h, unsigned& m) {
    h = wv_pk(a, b);
    m = wv_pk(a - __uint_as_float(h << 16), b - __uint_as_float(h & 0xFFFF0000u));   // (x - hi is exact)
}
__device__ __forceinline__ void wv_split64(const float (&x)[16], Op64& o) {
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned h, m;
            wv_split2(x[8 * q + 2 * p], x[8 * q + 2 * p + 1], h, m);
            o.h[q][p] = h; o.m[q][p] = m;
        }
}
__device__ __forceinline__ void wv_split16(float a, float b, float c, float d, Op16& o) {
    unsigned h0, m0, h1, m1;
    wv_split2(a, b, h0, m0);
    wv_split2(c, d, h1, m1);
    o.h = (wv_u32x2){h0, h1}; o.m = (wv_u32x2){m0, m1};
}

__device__ __forceinline__ f32x4 wv_mfma32(wv_u32x4 a, wv_u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wv_bf16x8, a), __builtin_bit_cast(wv_bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 wv_mfma16(wv_u32x2 a, wv_u32x2 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(wv_s16x4, a), __builtin_bit_cast(wv_s16x4, b), c, 0, 0, 0);
}
// C[m][n] += sum over 64 features of A[m][.] B[.][n]: the small cross terms first, then hi.hi
__device__ __forceinline__ f32x4 wv_mm64(const wv_u32x4 (&ah)[2], const wv_u32x4 (&am)[2], const Op64& b, f32x4 acc) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        acc = wv_mfma32(am[q], b.h[q], acc);
        acc = wv_mfma32(ah[q], b.m[q], acc);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) acc = wv_mfma32(ah[q], b.h[q], acc);
    return acc;
}
__device__ __forceinline__ f32x4 wv_mm64(const Op64& a, const Op64& b, f32x4 acc) { return wv_mm64(a.h, a.m, b, acc); }
__device__ __forceinline__ f32x4 wv_mm16(const Op16& a, const Op16& b, f32x4 acc) {
    acc = wv_mfma16(a.m, b.h, acc);
    acc = wv_mfma16(a.h, b.m, acc);
    return wv_mfma16(a.h, b.h, acc);
}

// ---- weight fragments (enc_wave_prep_k) -------------------------------------------------------------------------------------
struct WFrag { wv_u32x4 h[4][2], m[4][2]; };   // [strip][k step]
__device__ __forceinline__ void wv_wload(WFrag& w, const uint32_t* __restrict__ wf, int l, int mat, int orient, int lane) {
    const wv_u32x4* p = reinterpret_cast<const wv_u32x4*>(wf + ((size_t)(l * 6 + mat) * 2 + orient) * WV_FRAG_WORDS) + lane;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            w.h[s][q] = p[((s * 2 + q) * 2 + 0) * 64];
            w.m[s][q] = p[((s * 2 + q) * 2 + 1) * 64];
        }
}
// T(Y) = W-product of T(X): out[4 s + j] = Y[c][16 s + 4 g + j]   (weights as A: M = output feature, X as B: N = token)
__device__ __forceinline__ void wv_gemm_t(const WFrag& w, const Op64& x, float (&out)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f32x4 a = wv_mm64(w.h[s], w.m[s], x, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * s + j] = a[j];
    }
}
// F(Y): out[4 s + j] = Y[4 g + j][16 s + c]   (X as A: M = token, weights as B: N = output feature)
__device__ __forceinline__ void wv_gemm_f(const Op64& x, const WFrag& w, float (&out)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        Op64 b;
        b.h[0] = w.h[s][0]; b.h[1] = w.h[s][1]; b.m[0] = w.m[s][0]; b.m[1] = w.m[s][1];
        const f32x4 a = wv_mm64(x, b, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
        for (int j = 0; j < 4; ++j) out[4 * s + j] = a[j];
    }
}

// ---- tile <-> global, [16][64] fp32 row-major -----------------------------------------------------------------------------------
__device__ __forceinline__ void wv_ld_row(const float* row, int g, float (&x)[16]) {   // the lane's own row: x[4 s + j] = row[16 s + 4 g + j]
    const float* p = row + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(p + 16 * s);
        x[4 * s] = v.x; x[4 * s + 1] = v.y; x[4 * s + 2] = v.z; x[4 * s + 3] = v.w;
    }
}
__device__ __forceinline__ void wv_st_row(float* row, int g, const float (&x)[16]) {
    float* p = row + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<float4*>(p + 16 * s) = make_float4(x[4 * s], x[4 * s + 1], x[4 * s + 2], x[4 * s + 3]);
}
__device__ __forceinline__ void wv_ld_t(const float* base, int c, int g, float (&x)[16]) {
    const float* p = base + c * WV_D + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 v = *reinterpret_cast<const float4*>(p + 16 * s);
        x[4 * s] = v.x; x[4 * s + 1] = v.y; x[4 * s + 2] = v.z; x[4 * s + 3] = v.w;
    }
}
__device__ __forceinline__ void wv_st_t(float* base, int c, int g, const float (&x)[16]) {
    float* p = base + c * WV_D + 4 * g;
#pragma unroll
    for (int s = 0; s < 4; ++s) *reinterpret_cast<float4*>(p + 16 * s) = make_float4(x[4 * s], x[4 * s + 1], x[4 * s + 2], x[4 * s + 3]);
}
__device__ __forceinline__ void wv_ld_f(const float* base, int c, int g, float (&f)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 * s + j] = base[(4 * g + j) * WV_D + 16 * s + c];
}
__device__ __forceinline__ void wv_st_f(float* base, int c, int g, const float (&f)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) base[(4 * g + j) * WV_D + 16 * s + c] = f[4 * s + j];
}
// a parameter vector in the lane's T order: v[4 s + j] = par[16 s + 4 g + j]   (LDS)
__device__ __forceinline__ void wv_par_t(const float* par, int g, float (&v)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 t = *reinterpret_cast<const float4*>(par + 16 * s + 4 * g);
        v[4 * s] = t.x; v[4 * s + 1] = t.y; v[4 * s + 2] = t.z; v[4 * s + 3] = t.w;
    }
}

// ---- sums ---------------------------------------------------------------------------------------------------------------------------
// over the four lanes that share a token (lane ^ 16, lane ^ 32)
__device__ __forceinline__ float wv_gsum(float v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ float wv_gmax(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ int wv_gsum_i(int v) {
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
// Column sums of a T-layout tile over its 16 tokens: a halving butterfly over the 16 lanes of a row -- each step a lane keeps one
// half of its registers and adds the partner's copy of that half (15 adds + selects instead of 16 four-step reductions).  Lane c ends
// with the sum of register r(c) = 8 b3 + 4 b2 + 2 b0 + b1 (b_k = bit k of c), i.e. of feature 16 (r >> 2) + 4 g + (r & 3).
#define WV_DPP_ROR8 0x128
__device__ __forceinline__ float wv_colsum(const float (&x)[16], int c) {
    const bool b3 = (c & 8) != 0, b2 = (c & 4) != 0, b0 = (c & 1) != 0, b1 = (c & 2) != 0;
    float y8[8], y4[4], y2[2];
#pragma unroll
    for (int r = 0; r < 8; ++r) y8[r] = (b3 ? x[8 + r] : x[r]) + se_dpp<WV_DPP_ROR8>(b3 ? x[r] : x[8 + r]);               // partner c ^ 8
#pragma unroll
    for (int r = 0; r < 4; ++r) y4[r] = (b2 ? y8[4 + r] : y8[r]) + se_dpp<SE_DPP_HALF_MIRROR>(b2 ? y8[r] : y8[4 + r]);    // partner c ^ 7
#pragma unroll
    for (int r = 0; r < 2; ++r) y2[r] = (b0 ? y4[2 + r] : y4[r]) + se_dpp<SE_DPP_XOR1>(b0 ? y4[r] : y4[2 + r]);           // partner c ^ 1
    return (b1 ? y2[1] : y2[0]) + se_dpp<SE_DPP_XOR2>(b1 ? y2[0] : y2[1]);                                                // partner c ^ 2
}
// lane -> feature of its column sum, and back
__device__ __forceinline__ int wv_colsum_feature(int lane) {
    const int c = lane & 15, g = lane >> 4;
    const int r = ((c & 8) ? 8 : 0) + ((c & 4) ? 4 : 0) + ((c & 1) ? 2 : 0) + ((c & 2) ? 1 : 0);
    return 16 * (r >> 2) + 4 * g + (r & 3);
}
__device__ __forceinline__ int wv_colsum_lane(int f) {
    const int s = f >> 4, g = (f >> 2) & 3, j = f & 3, r = 4 * s + j;
    const int c = ((r & 8) ? 8 : 0) + ((r & 4) ? 4 : 0) + ((r & 2) ? 1 : 0) + ((r & 1) ? 2 : 0);
    return 16 * g + c;
}

// LayerNorm of the lane's token (eps 1e-8, biased variance: nn.LayerNorm, SASRec/main.py:89,94,106); xh = normalised input
__device__ __forceinline__ void wv_ln_stats(const float (&x)[16], float& mean, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    mean = wv_gsum(s) * (1.0f / WV_D);
    float q2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) { const float d = x[i] - mean; q2 = fmaf(d, d, q2); }
    rstd = 1.0f / sqrtf(wv_gsum(q2) * (1.0f / WV_D) + 1e-8f);
}
// dx = rstd (d - mean(d) - xh mean(d xh)), d = dy gamma
__device__ __forceinline__ void wv_ln_bwd(const float (&dy)[16], const float (&xh)[16], const float (&gamma)[16], float rstd, float (&dx)[16]) {
    float d[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        d[i] = dy[i] * gamma[i];
        s1 += d[i];
        s2 = fmaf(d[i], xh[i], s2);
    }
    s1 = wv_gsum(s1) * (1.0f / WV_D);
    s2 = wv_gsum(s2) * (1.0f / WV_D);
#pragma unroll
    for (int i = 0; i < 16; ++i) dx[i] = rstd * (d[i] - s1 - xh[i] * s2);
}

// 16 x 16 transpose Rt -> R (or back) through the wave's LDS scratch: in[j] = M[c][4 g + j] -> out[j] = M[4 g + j][c]
__device__ __forceinline__ void wv_tr16(float* scr, int c, int g, const f32x4& in, f32x4& out) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (earlier readers of the scratch are done: one wave, in order)
    *reinterpret_cast<f32x4*>(scr + c * 20 + 4 * g) = in;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 4; ++j) out[j] = scr[(4 * g + j) * 20 + c];
}

// ---- weight preparation: fp32 [64][64] -> bf16 hi / mid fragment planes in the kernel's k order, both orientations ---------------------
__global__ __launch_bounds__(256) void enc_wave_prep_k(SasrecParams P, int L, uint32_t* __restrict__ wf) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int lane = t & 63, q = (t >> 6) & 1, s = (t >> 7) & 3, o = (t >> 9) & 1, lm = t >> 10;
    if (lm >= 6 * L) return;
    const int l = lm / 6, m = lm % 6, c = lane & 15, g = lane >> 4;
    const SasrecBlockParams& W = P.blk[l];
    const float* w = (m < 3) ? W.in_w + m * WV_D * WV_D : (m == 3) ? W.out_w : (m == 4) ? W.w1 : W.w2;
    unsigned h[4], md[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 2 * p + e;
            const int k = 16 * (2 * q + (i >> 2)) + 4 * g + (i & 3);
            v[e] = (o == 0) ? w[(16 * s + c) * WV_D + k] : w[k * WV_D + 16 * s + c];
        }
        wv_split2(v[0], v[1], h[p], md[p]);
    }
    wv_u32x4* dst = reinterpret_cast<wv_u32x4*>(wf + ((size_t)lm * 2 + o) * WV_FRAG_WORDS) + lane;
    dst[((s * 2 + q) * 2 + 0) * 64] = (wv_u32x4){h[0], h[1], h[2], h[3]};
    dst[((s * 2 + q) * 2 + 1) * 64] = (wv_u32x4){md[0], md[1], md[2], md[3]};
}

// Phase stamps (diagnostic build only: `make -C recboard_amd/csrc encprof`, scripts/wave_phases.py): lane 0 of wave 0 of workgroup 0
// records the shader clock at the phase boundaries of its first item's tile.
#ifdef WV_PROFILE
#define WV_MARKS 96
__device__ unsigned long long g_wave_marks[WV_MARKS];
extern "C" int re_dbg_enc_marks_wave(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_marks), sizeof(unsigned long long) * WV_MARKS) == hipSuccess ? 0 : 1;
}
#define WV_MARK() do { if (blockIdx.x == 0 && tid == 0 && k == 0 && mk < WV_MARKS) g_wave_marks[mk] = __builtin_amdgcn_s_memtime(); ++mk; } while (0)
#else
#define WV_MARK() do { } while (0)
#endif

// LDS carve-up (floats): parameters | per-wave vector-gradient stage | dK / dV exchange slots | transpose scratch | loss partials
__host__ __device__ inline size_t wv_lds_floats(int L) {
    return (size_t)(WV_NPAR * L + 2) * WV_D + (size_t)WV_NW * L * EG_NVEC * WV_D + (size_t)WV_XCH_SLOTS * 1024 + WV_NW * WV_TR + 16;
}

__global__ __launch_bounds__(WV_NT) void enc_wave_step_k(SeEmbed em, const int64_t* __restrict__ seq, int B, int S, int L, SasrecParams P,
                                                         float drop_scale, uint32_t thresh, uint32_t seed, float* __restrict__ u,
                                                         float* __restrict__ tape, EncTape T, const void* __restrict__ planp, EncHead H,
                                                         float* __restrict__ dOut, float* __restrict__ gtape, float* __restrict__ slab,
                                                         const uint32_t* __restrict__ seed_dev, float emb_scale,
                                                         const uint32_t* __restrict__ wf) {
    if (seed_dev) seed ^= seed_dev[0];
    extern __shared__ __align__(16) float lds[];
    const EncPlan PL = enc_plan_view(planp, B, S);
    const int n_items = PL.hdr[0];
    if (PL.hdr[5] > 0) {   // a plan with split sequences (kinds 2 / 3) is not for this kernel: say so instead of computing garbage
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            __hip_atomic_store(reinterpret_cast<unsigned*>(tape + T.off_FLAGS) + enc_plan_max_tiles(B, S) * EP_FLAG_WORDS, 2u, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            H.loss[0] = __builtin_nanf("");
        }
        return;
    }
    float* s_par = lds;
    float* s_acc = s_par + (WV_NPAR * L + 2) * WV_D;
    float* s_xch = s_acc + WV_NW * L * EG_NVEC * WV_D;
    float* s_tr = s_xch + WV_XCH_SLOTS * 1024;
    float* s_red = s_tr + WV_NW * WV_TR;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const float inv_sqrt_d = 0.125f;
    const int64_t NR = 16 * enc_plan_max_tiles(B, S);
    float* scr = s_tr + wave * WV_TR;
    float* acc_w = s_acc + (size_t)wave * L * EG_NVEC * WV_D;

    // the small parameters of every block and lastLN: once per workgroup
    for (int e = tid; e < (WV_NPAR * L + 2) * WV_D; e += WV_NT) {
        const int v = e / WV_D, cc = e % WV_D;
        const float* p;
        if (v >= WV_NPAR * L) p = (v == WV_NPAR * L) ? P.last_w : P.last_b;
        else {
            const SasrecBlockParams& W = P.blk[v / WV_NPAR];
            const int k = v % WV_NPAR;
            p = (k == 0) ? W.ln_a_w : (k == 1) ? W.ln_a_b : (k < 5) ? W.in_b + (k - 2) * WV_D : (k == 5) ? W.out_b : (k == 6) ? W.ln_f_w
              : (k == 7) ? W.ln_f_b : (k == 8) ? W.b1 : W.b2;
        }
        s_par[e] = p[cc];
    }
    __syncthreads();

    for (int k = 0; k * (int)gridDim.x < n_items; ++k) {
        const int wi = enc_item_of(k, blockIdx.x, gridDim.x);
        if (wi >= n_items) continue;                      // (workgroup-uniform)
        const EncItem it = enc_item(PL, wi);
        const bool multi = it.kind == 1 && it.nt > 1;     // tiles of ONE sequence: k, v and dK, dV cross waves
        const int tt = wave;                              // this wave's tile of the item
        const bool active = tt < it.nt;
        float head_loss = 0.f;
        if (!active) {
            // no tile: zero vector-gradient stage, and the barriers of a multi-tile item (one per block forward, four per block backward)
            for (int e = lane; e < L * EG_NVEC * WV_D; e += 64) acc_w[e] = 0.f;
            if (multi)
                for (int b = 0; b < 5 * L; ++b) __syncthreads();
        } else {
        const int64_t row0 = (int64_t)(it.tile0 + tt) * 16;          // compact row of the tile's first row
        const int64_t irow0 = (int64_t)it.tile0 * 16;                // ... of the item's
        const int klo = multi ? 0 : tt;                              // key tiles klo .. tt
        // ---- the lane's token: which (sequence, position) it is, its sequence's first row inside the item, the pads in front
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
        // ---- x0 = E[item] sqrt(D) + P[position], dropout (SASRec/main.py:181-187)
        float x[16];
        unsigned emask = 0xFFFFu;
        {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = 0.f;
            if (real) {
                float e[16], pp[16];
                wv_ld_row(em.E + item * WV_D, g, e);
                wv_ld_row(em.P + (int64_t)(gid % S) * WV_D, g, pp);
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fmaf(e[i], em.scale, pp[i]);
                if (thresh) {
                    emask = 0u;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const uint32_t el = (uint32_t)((int64_t)gid * WV_D + 16 * (i >> 2) + 4 * g + (i & 3));
                        const bool kp = re_keep(seed, RE_STREAM_EMBED, el, thresh);
                        emask |= (kp ? 1u : 0u) << i;
                        x[i] = kp ? x[i] * drop_scale : 0.f;
                    }
                }
            }
        }
        // =================================================== forward ===================================================
        // Weight fragments: two register sets, each requested one product ahead of its use (a request issued right in front of its
        // product exposes a whole L2 round trip per product: 26 of them were half of the launch).
        WFrag wa, wb;
        int mk = 0; (void)mk;
        WV_MARK();
        wv_wload(wa, wf, 0, 0, 0, lane);                  // block 0's Wq
        // the loss head's indices: requested now, used after the last block
        int64_t hpr = 0, hng = 0;
        if (gid >= 0) { hpr = H.pos[gid] + H.e_off; hng = H.neg[gid] + H.e_off; }
        const float hgs = 1.0f / (float)H.count[0];
        for (int l = 0; l < L; ++l) {
            float* tp = tape + (int64_t)l * T.per_block;
            const float* par = s_par + l * WV_NPAR * WV_D;
            float pv[16], a[16];
            // ---- a = LN_a(x)
            {
                float mean, rstd, gw[16];
                wv_ln_stats(x, mean, rstd);
                wv_par_t(par + 0 * WV_D, g, gw);
                wv_par_t(par + 1 * WV_D, g, pv);
#pragma unroll
                for (int i = 0; i < 16; ++i) a[i] = fmaf((x[i] - mean) * rstd, gw[i], pv[i]);
            }
            wv_wload(wb, wf, l, 1, 0, lane);              // Wk
            wv_st_t(tp + T.off_X + row0 * WV_D, c, g, x);
            wv_st_t(tp + T.off_A + row0 * WV_D, c, g, a);
            // ---- q = a Wq^T + bq, k = x Wk^T + bk (T), v = x Wv^T + bv (F)
            Op64 ao, xo;
            wv_split64(a, ao);
            wv_split64(x, xo);
            float q[16], kk[16], vf[16];
            WV_MARK();
            wv_gemm_t(wa, ao, q);
            wv_wload(wa, wf, l, 2, 0, lane);              // Wv
            wv_par_t(par + 2 * WV_D, g, pv);
#pragma unroll
            for (int i = 0; i < 16; ++i) q[i] += pv[i];
            wv_gemm_t(wb, xo, kk);
            wv_wload(wb, wf, l, 3, 0, lane);              // Wo
            float bk[16];
            wv_par_t(par + 3 * WV_D, g, bk);
#pragma unroll
            for (int i = 0; i < 16; ++i) kk[i] += bk[i];
            wv_gemm_f(xo, wa, vf);
            wv_wload(wa, wf, l, 4, 0, lane);              // W1
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float bvs = par[4 * WV_D + 16 * s + c];
#pragma unroll
                for (int j = 0; j < 4; ++j) vf[4 * s + j] += bvs;
            }
            wv_st_t(tp + T.off_Q + row0 * WV_D, c, g, q);
            wv_st_t(tp + T.off_K + row0 * WV_D, c, g, kk);
            wv_st_f(tp + T.off_V + row0 * WV_D, c, g, vf);
            if (multi) __syncthreads();                   // (with vmcnt(0): the item's k, v of this block are in L2 for its other waves)
            WV_MARK();
            // ---- scores Rt(S)[kt] = q k^T / sqrt(D) over the key tiles
            Op64 qo;
            wv_split64(q, qo);
            f32x4 p[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                p[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kt < klo || kt > tt) continue;         // (wave-uniform)
                Op64 ko;
                if (kt == tt) wv_split64(kk, ko);
                else {
                    float t[16];
                    wv_ld_t(tp + T.off_K + (irow0 + 16 * kt) * WV_D, c, g, t);
                    wv_split64(t, ko);
                }
                p[kt] = wv_mm64(ko, qo, p[kt]);
#pragma unroll
                for (int j = 0; j < 4; ++j) p[kt][j] *= inv_sqrt_d;
            }
            // ---- softmax over the keys of the token's own sequence up to itself, plus the virtual pad key (multiplicity n_out,
            //      score q.b_k / sqrt(D), value b_v), dropout on the probabilities -- enc_fwd_item.h, same arithmetic
            const unsigned span = gid >= 0 ? (unsigned)(io - st) : 0u;
            float mx = -INFINITY;
            unsigned okm = 0u;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int jo = 16 * kt + 4 * g + j;
                    const bool ok = kt >= klo && kt <= tt && gid >= 0 && (unsigned)(jo - st) <= span;
                    okm |= (ok ? 1u : 0u) << (4 * kt + j);
                    mx = fmaxf(mx, ok ? p[kt][j] : -INFINITY);
                }
            float dq = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) dq = fmaf(q[i], bk[i], dq);
            dq = wv_gsum(dq);
            const float spad = (gid >= 0 && n_out > 0) ? dq * inv_sqrt_d : -INFINITY;
            mx = wv_gmax(fmaxf(mx, spad));
            float sum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    p[kt][j] = ((okm >> (4 * kt + j)) & 1u) ? expf(p[kt][j] - mx) : 0.f;
                    sum += p[kt][j];
                }
            sum = wv_gsum(sum);
            const float epad = (spad == -INFINITY) ? 0.f : expf(spad - mx);
            sum += (float)n_out * epad;
            const float inv = (gid >= 0) ? 1.0f / sum : 0.f;
            const float ppad = epad * inv;
            float kept = (float)n_out;
            if (thresh) {   // each of the n_out pad keys has its own dropout bit (element (b, s_i, jj)): the token's four lanes share them
                int cnt = 0;
                const uint32_t e0 = (uint32_t)((int64_t)gid * S);
                for (int jj = g; jj < S; jj += 4) cnt += (jj < n_out && gid >= 0 && re_keep(seed, RE_STREAM_ATTN(l), e0 + (uint32_t)jj, thresh)) ? 1 : 0;
                cnt = wv_gsum_i(cnt);
                kept = (float)cnt * drop_scale;
            }
            const float wvv = (gid >= 0) ? ppad * kept : 0.f;
            if (g == 0) *reinterpret_cast<float2*>(tp + T.off_PP + (row0 + c) * 2) = make_float2(ppad, wvv);
            unsigned amask = 0xFFFFu;
            f32x4 pd[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) p[kt][j] *= inv;
                pd[kt] = p[kt];
                if (kt >= klo && kt <= tt)   // pre-dropout probabilities (0 outside the token's window): row c, key columns 16 kt + 4 g ..
                    *reinterpret_cast<f32x4*>(tp + T.off_P + (row0 + c) * EP_PW + 16 * kt + 4 * g) = p[kt];
            }
            if (thresh) {
                amask = 0u;
                const uint32_t e0 = (uint32_t)((int64_t)gid * S + n_out - st);   // + item-local key row: the key's position in the sequence
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    if (kt < klo || kt > tt) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const bool kp = re_keep(seed, RE_STREAM_ATTN(l), e0 + (uint32_t)(16 * kt + 4 * g + j), thresh);
                        amask |= (kp ? 1u : 0u) << (4 * kt + j);
                        pd[kt][j] = kp ? p[kt][j] * drop_scale : 0.f;
                    }
                }
            }
            WV_MARK();
            // ---- o = Pd v + w b_v   (T(o) = sum over key tiles of F(v)-as-A x Rt(Pd)-as-B)
            float o[16];
            {
                f32x4 oa[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) oa[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) {
                    if (kt < klo || kt > tt) continue;
                    Op16 po;
                    wv_split16(pd[kt][0], pd[kt][1], pd[kt][2], pd[kt][3], po);
                    float vt[16];
                    if (kt == tt) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) vt[i] = vf[i];
                    } else {
                        wv_ld_f(tp + T.off_V + (irow0 + 16 * kt) * WV_D, c, g, vt);
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        Op16 vo;
                        wv_split16(vt[4 * s], vt[4 * s + 1], vt[4 * s + 2], vt[4 * s + 3], vo);
                        oa[s] = wv_mm16(vo, po, oa[s]);
                    }
                }
                wv_par_t(par + 4 * WV_D, g, pv);   // b_v in T order
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[4 * s + j] = fmaf(wvv, pv[4 * s + j], oa[s][j]);
            }
            wv_st_t(tp + T.off_O + row0 * WV_D, c, g, o);
            // ---- x1 = o Wo^T + bo + x
            float x1[16];
            {
                Op64 oo;
                wv_split64(o, oo);
                WV_MARK();
                wv_gemm_t(wb, oo, x1);
                wv_wload(wb, wf, l, 5, 0, lane);          // W2
                wv_par_t(par + 5 * WV_D, g, pv);
#pragma unroll
                for (int i = 0; i < 16; ++i) x1[i] += pv[i] + x[i];
            }
            wv_st_t(tp + T.off_X1 + row0 * WV_D, c, g, x1);
            // ---- y = LN_f(x1)
            float y[16];
            {
                float mean, rstd, gw[16];
                wv_ln_stats(x1, mean, rstd);
                wv_par_t(par + 6 * WV_D, g, gw);
                wv_par_t(par + 7 * WV_D, g, pv);
#pragma unroll
                for (int i = 0; i < 16; ++i) y[i] = fmaf((x1[i] - mean) * rstd, gw[i], pv[i]);
            }
            wv_st_t(tp + T.off_Y + row0 * WV_D, c, g, y);
            // ---- hr = relu(dropout1(y W1^T + b1))
            float hr[16];
            unsigned hmask = 0u;
            {
                Op64 yo;
                wv_split64(y, yo);
                WV_MARK();
                wv_gemm_t(wa, yo, hr);
                if (l + 1 < L) wv_wload(wa, wf, l + 1, 0, 0, lane);   // the next block's Wq,
                else wv_wload(wa, wf, L - 1, 5, 1, lane);            // or the backward's first: W2 of the last block
                wv_par_t(par + 8 * WV_D, g, pv);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = hr[i] + pv[i];
                    if (thresh) {
                        const uint32_t el = (uint32_t)((int64_t)gid * WV_D + 16 * (i >> 2) + 4 * g + (i & 3));
                        v = re_keep(seed, RE_STREAM_FFN1(l), el, thresh) ? v * drop_scale : 0.f;
                    }
                    hr[i] = fmaxf(v, 0.f);
                    hmask |= (hr[i] > 0.f ? 1u : 0u) << i;
                }
            }
            wv_st_t(tp + T.off_HR + row0 * WV_D, c, g, hr);
            // ---- x' = dropout2(hr W2^T + b2) + y, pad rows zeroed
            unsigned m2 = 0xFFFFu;
            {
                Op64 ho;
                wv_split64(hr, ho);
                float z[16];
                WV_MARK();
                wv_gemm_t(wb, ho, z);
                wv_par_t(par + 9 * WV_D, g, pv);
                if (thresh) m2 = 0u;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    float v = z[i] + pv[i];
                    if (thresh) {
                        const uint32_t el = (uint32_t)((int64_t)gid * WV_D + 16 * (i >> 2) + 4 * g + (i & 3));
                        const bool kp = re_keep(seed, RE_STREAM_FFN2(l), el, thresh);
                        m2 |= (kp ? 1u : 0u) << i;
                        v = kp ? v * drop_scale : 0.f;
                    }
                    x[i] = dead ? 0.f : v + y[i];
                }
            }
            // the block's mask bits for the backward: one word per lane ([tile][4][64] words of the tape's mask array)
            {
                uint32_t* mkw = reinterpret_cast<uint32_t*>(tp + T.off_MK) + row0 * 16;
                mkw[lane] = m2 | (hmask << 16);
                mkw[64 + lane] = amask;
            }
        }
        WV_MARK();
        // ---- u = LN_last(x_L)
        float mean_l, rstd_l, uu[16], xh[16], glw[16];
        {
            float bb[16];
            wv_ln_stats(x, mean_l, rstd_l);
            wv_par_t(s_par + WV_NPAR * L * WV_D, g, glw);
            wv_par_t(s_par + (WV_NPAR * L + 1) * WV_D, g, bb);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                xh[i] = (x[i] - mean_l) * rstd_l;
                uu[i] = fmaf(xh[i], glw[i], bb[i]);
            }
        }
        if (gid >= 0) wv_st_row(u + (int64_t)gid * WV_D, g, uu);
        // ---- loss head (SASRec/main.py:199-215): pl = <u, E[pos]>, nl = <u, E[neg]>; the rows' gradient contributions and keys
        float du[16];
        {
            int64_t pr = hpr, ng = hng;
            const bool realh = item > 0 && item < H.R;
            const bool ok = realh && pr > 0 && pr < H.R && ng > 0 && ng < H.R;
            if (!ok) { pr = 0; ng = 0; }
            float ep[16], en[16];
            wv_ld_row(H.E + pr * WV_D, g, ep);
            wv_ld_row(H.E + ng * WV_D, g, en);
            float pl = 0.f, nl = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { pl = fmaf(uu[i], ep[i], pl); nl = fmaf(uu[i], en[i], nl); }
            pl = wv_gsum(pl);
            nl = wv_gsum(nl);
            const float gs = hgs;
            float dpl, dnl;
            if (H.kind == RE_LOSS_BCE) { dpl = -re_sigmoid(-pl) * gs; dnl = re_sigmoid(nl) * gs; }
            else { const float sg = re_sigmoid(nl - pl) * gs; dpl = -sg; dnl = sg; }
            if (!ok) { dpl = 0.f; dnl = 0.f; }
            float gp[16], gn[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                du[i] = fmaf(dpl, ep[i], dnl * en[i]);
                gp[i] = dpl * uu[i];
                gn[i] = dnl * uu[i];
            }
            const int64_t r = row0;
            wv_st_t(H.dU_rows + r * WV_D, c, g, du);
            if (ok) {
                wv_st_t(H.g_rows + (NR + r) * WV_D, c, g, gp);
                wv_st_t(H.g_rows + (2 * NR + r) * WV_D, c, g, gn);
            }
            if (g == 0) {
                H.keys[r + c] = realh ? (int)item : 0;
                H.keys[NR + r + c] = (int)pr;
                H.keys[2 * NR + r + c] = (int)ng;
                if (ok) head_loss = (H.kind == RE_LOSS_BCE) ? re_softplus(-pl) + re_softplus(nl) : re_softplus(nl - pl);
            }
        }
        // =================================================== backward ===================================================
        WV_MARK();
        float dx[16];
        {
            // lastLN: dgamma, dbeta, dx_L
            float t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = du[i] * xh[i];
            float* ac = acc_w + (size_t)(L - 1) * EG_NVEC * WV_D;
            ac[10 * WV_D + lane] = wv_colsum(t, c);
            ac[11 * WV_D + lane] = wv_colsum(du, c);
            wv_ln_bwd(du, xh, glw, rstd_l, dx);
        }
        for (int l = L - 1; l >= 0; --l) {
            const float* tp = tape + (int64_t)l * T.per_block;
            const float* par = s_par + l * WV_NPAR * WV_D;
            float* gp = gtape + (int64_t)l * EG_NMAT * NR * WV_D + row0 * WV_D;
            float* ac = acc_w + (size_t)l * EG_NVEC * WV_D;
            if (l != L - 1) { ac[10 * WV_D + lane] = 0.f; ac[11 * WV_D + lane] = 0.f; }
            const uint32_t* mkw = reinterpret_cast<const uint32_t*>(tp + T.off_MK) + row0 * 16;
            const unsigned mw = mkw[lane], amask = mkw[64 + lane];
            float pv[16], t1[16];
            // ---- pad mask of the block output, dO2 = dX' * dropout2 mask
            float dz[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                dx[i] = dead ? 0.f : dx[i];
                dz[i] = !thresh ? dx[i] : ((mw >> i) & 1u) ? dx[i] * drop_scale : 0.f;
            }
            wv_wload(wb, wf, l, 4, 1, lane);              // W1 (wa holds W2)
            // this block's tape rows for the phases below: requested now
            float x1[16], vt_own[16];
            wv_ld_t(tp + T.off_X1 + row0 * WV_D, c, g, x1);
            wv_ld_t(tp + T.off_V + row0 * WV_D, c, g, vt_own);
            const float2 ppw = *reinterpret_cast<const float2*>(tp + T.off_PP + (row0 + c) * 2);
            wv_st_t(gp + 0 * NR * WV_D, c, g, dz);
            ac[5 * WV_D + lane] = wv_colsum(dz, c);
            // ---- A. dH = (dO2 W2) * (hr > 0) * scale
            float dh[16];
            {
                Op64 o;
                wv_split64(dz, o);
                WV_MARK();
                wv_gemm_t(wa, o, dh);
                wv_wload(wa, wf, l, 3, 1, lane);          // Wo
#pragma unroll
                for (int i = 0; i < 16; ++i) dh[i] = ((mw >> (16 + i)) & 1u) ? dh[i] * drop_scale : 0.f;
            }
            wv_st_t(gp + 1 * NR * WV_D, c, g, dh);
            ac[4 * WV_D + lane] = wv_colsum(dh, c);
            // ---- B. dY = dH W1 + dX'
            float dy[16];
            {
                Op64 o;
                wv_split64(dh, o);
                WV_MARK();
                wv_gemm_t(wb, o, dy);
                wv_wload(wb, wf, l, 0, 1, lane);          // Wq
#pragma unroll
                for (int i = 0; i < 16; ++i) dy[i] += dx[i];
            }
            // ---- C. LN_f backward: dgamma_f, dbeta_f, dX1
            float dx1[16];
            {
                float mean, rstd;
                WV_MARK();
                wv_ln_stats(x1, mean, rstd);
#pragma unroll
                for (int i = 0; i < 16; ++i) { x1[i] = (x1[i] - mean) * rstd; t1[i] = dy[i] * x1[i]; }
                ac[8 * WV_D + lane] = wv_colsum(t1, c);
                ac[9 * WV_D + lane] = wv_colsum(dy, c);
                wv_par_t(par + 6 * WV_D, g, pv);
                wv_ln_bwd(dy, x1, pv, rstd, dx1);
            }
            wv_st_t(gp + 2 * NR * WV_D, c, g, dx1);
            ac[3 * WV_D + lane] = wv_colsum(dx1, c);
            // ---- D. dO = dX1 Wo, in both layouts (the same fragments, operands swapped)
            float dO[16], dOf[16];
            {
                Op64 o;
                wv_split64(dx1, o);
                WV_MARK();
                wv_gemm_t(wa, o, dO);
                wv_gemm_f(o, wa, dOf);
                wv_wload(wa, wf, l, 1, 1, lane);          // Wk
            }
            // ---- E. attention backward (enc_bwd_item.h, same arithmetic)
            WV_MARK();
            const float ppad = ppw.x, wvv = ppw.y;
            // q and the own tile's k in F layout, x for LN_a's backward: requested here, used behind the score-type products
            float qf[16], kf_own[16], xx[16];
            wv_ld_f(tp + T.off_Q + row0 * WV_D, c, g, qf);
            wv_ld_f(tp + T.off_K + row0 * WV_D, c, g, kf_own);
            wv_ld_t(tp + T.off_X + row0 * WV_D, c, g, xx);
            float bvt[16], tdot = 0.f;
            wv_par_t(par + 4 * WV_D, g, bvt);
#pragma unroll
            for (int i = 0; i < 16; ++i) { tdot = fmaf(dO[i], bvt[i], tdot); t1[i] = wvv * dO[i]; }
            tdot = wv_gsum(tdot);
            float acc_bv = wv_colsum(t1, c);   // d b_v through the virtual pad key: sum_i w_i dO_i
            Op64 doo;
            wv_split64(dO, doo);
            f32x4 p[4], pd[4], dp[4];
            float srow = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                p[kt] = pd[kt] = dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kt < klo || kt > tt) continue;
                p[kt] = *reinterpret_cast<const f32x4*>(tp + T.off_P + (row0 + c) * EP_PW + 16 * kt + 4 * g);
                float vt[16];
                if (kt == tt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) vt[i] = vt_own[i];
                } else {
                    wv_ld_t(tp + T.off_V + (irow0 + 16 * kt) * WV_D, c, g, vt);
                }
                Op64 vo;
                wv_split64(vt, vo);
                const f32x4 raw = wv_mm64(vo, doo, (f32x4){0.f, 0.f, 0.f, 0.f});   // (dO_i . v_j) for token i = c, keys 4 g + j
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float mf = !thresh ? 1.0f : ((amask >> (4 * kt + j)) & 1u) ? drop_scale : 0.f;
                    pd[kt][j] = p[kt][j] * mf;
                    dp[kt][j] = (p[kt][j] != 0.f) ? raw[j] * mf : 0.f;
                    srow = fmaf(dp[kt][j], p[kt][j], srow);
                }
            }
            srow = wv_gsum(srow);
            srow = fmaf(tdot, wvv, srow);                                           // the row dot includes the pad copies
            const float cpad = (wvv * tdot - (float)n_out * ppad * srow) * inv_sqrt_d;   // sum of dS over the pad copies
            f32x4 ds[4];
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int j = 0; j < 4; ++j) ds[kt][j] = p[kt][j] * (dp[kt][j] - srow) * inv_sqrt_d;
            // dQ = dS K + dS_pad b_k;  per key tile dV_kt = Pd^T dO, dK_kt = dS^T Q
            WV_MARK();
            Op16 qo4[4], do4[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                wv_split16(qf[4 * s], qf[4 * s + 1], qf[4 * s + 2], qf[4 * s + 3], qo4[s]);
                wv_split16(dOf[4 * s], dOf[4 * s + 1], dOf[4 * s + 2], dOf[4 * s + 3], do4[s]);
            }
            f32x4 dqa[4];
            float dk[16], dv[16];
#pragma unroll
            for (int s = 0; s < 4; ++s) dqa[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 16; ++i) { dk[i] = 0.f; dv[i] = 0.f; }
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                if (kt < klo || kt > tt) continue;
                float kf[16];
                if (kt == tt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) kf[i] = kf_own[i];
                } else {
                    wv_ld_f(tp + T.off_K + (irow0 + 16 * kt) * WV_D, c, g, kf);
                }
                Op16 dso, pdr, dsr;
                wv_split16(ds[kt][0], ds[kt][1], ds[kt][2], ds[kt][3], dso);
                f32x4 tr;
                wv_tr16(scr, c, g, pd[kt], tr);
                wv_split16(tr[0], tr[1], tr[2], tr[3], pdr);
                wv_tr16(scr, c, g, ds[kt], tr);
                wv_split16(tr[0], tr[1], tr[2], tr[3], dsr);
                float pk[16], pvv[16];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    Op16 ko;
                    wv_split16(kf[4 * s], kf[4 * s + 1], kf[4 * s + 2], kf[4 * s + 3], ko);
                    dqa[s] = wv_mm16(ko, dso, dqa[s]);                                          // T(dq): features x queries
                    const f32x4 a = wv_mm16(do4[s], pdr, (f32x4){0.f, 0.f, 0.f, 0.f});         // T(dv_kt): features x keys
                    const f32x4 b = wv_mm16(qo4[s], dsr, (f32x4){0.f, 0.f, 0.f, 0.f});         // T(dk_kt)
#pragma unroll
                    for (int j = 0; j < 4; ++j) { pvv[4 * s + j] = a[j]; pk[4 * s + j] = b[j]; }
                }
                if (kt == tt) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) { dv[i] = pvv[i]; dk[i] = pk[i]; }
                } else {
                    // this tile's contribution to an EARLIER tile's dV: slot (tt, kt), read by its owner behind the barrier below
                    // (the dK contribution follows in a second round through the same slots -- recomputed there, 12 MFMAs, rather
                    // than held in registers across the barriers)
                    float* sl = s_xch + (tt * (tt - 1) / 2 + kt) * 1024;
#pragma unroll
                    for (int i = 0; i < 16; ++i) sl[i * 64 + lane] = pvv[i];
                }
            }
            float dqv[16];
            float bk[16];
            wv_par_t(par + 3 * WV_D, g, bk);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int j = 0; j < 4; ++j) dqv[4 * s + j] = fmaf(cpad, bk[4 * s + j], dqa[s][j]);
            if (multi) {
                // round 1: dV partials are in the slots -> owners add them in tile order
                __syncthreads();
                for (int t = tt + 1; t < it.nt; ++t) {
                    const float* sl = s_xch + (t * (t - 1) / 2 + tt) * 1024;
#pragma unroll
                    for (int i = 0; i < 16; ++i) dv[i] += sl[i * 64 + lane];
                }
                __syncthreads();
                // round 2: dK partials (recomputed per earlier key tile: cheap, and no registers held across the barrier)
#pragma unroll
                for (int kt = 0; kt < 3; ++kt) {
                    if (kt >= tt) continue;
                    f32x4 tr;
                    Op16 dsr;
                    wv_tr16(scr, c, g, ds[kt], tr);
                    wv_split16(tr[0], tr[1], tr[2], tr[3], dsr);
                    float* sl = s_xch + (tt * (tt - 1) / 2 + kt) * 1024;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const f32x4 b = wv_mm16(qo4[s], dsr, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
                        for (int j = 0; j < 4; ++j) sl[(4 * s + j) * 64 + lane] = b[j];
                    }
                }
                __syncthreads();
                for (int t = tt + 1; t < it.nt; ++t) {
                    const float* sl = s_xch + (t * (t - 1) / 2 + tt) * 1024;
#pragma unroll
                    for (int i = 0; i < 16; ++i) dk[i] += sl[i * 64 + lane];
                }
                __syncthreads();
            }
            wv_st_t(gp + 3 * NR * WV_D, c, g, dqv);
            wv_st_t(gp + 4 * NR * WV_D, c, g, dk);
            wv_st_t(gp + 5 * NR * WV_D, c, g, dv);
            ac[0 * WV_D + lane] = wv_colsum(dqv, c);
            {
                float qt[16];
                wv_ld_t(tp + T.off_Q + row0 * WV_D, c, g, qt);
#pragma unroll
                for (int i = 0; i < 16; ++i) qt[i] *= cpad;
                ac[1 * WV_D + lane] = wv_colsum(qt, c) + wv_colsum(dk, c);   // d b_k: sum_i dS_pad_i q_i + the key rows
            }
            ac[2 * WV_D + lane] = acc_bv + wv_colsum(dv, c);
            // ---- G. dA = dQ Wq;  dX = dX1 + dK Wk + dV Wv + LN_a'(dA)
            float da[16];
            {
                Op64 o;
                wv_split64(dqv, o);
                WV_MARK();
                wv_gemm_t(wb, o, da);
                wv_wload(wb, wf, l, 2, 1, lane);          // Wv
                wv_split64(dk, o);
                wv_gemm_t(wa, o, t1);
                if (l > 0) wv_wload(wa, wf, l - 1, 5, 1, lane);   // the next block's W2
#pragma unroll
                for (int i = 0; i < 16; ++i) dx1[i] += t1[i];
                wv_split64(dv, o);
                wv_gemm_t(wb, o, t1);
#pragma unroll
                for (int i = 0; i < 16; ++i) dx1[i] += t1[i];
            }
            {
                float mean, rstd;
                WV_MARK();
                wv_ln_stats(xx, mean, rstd);
#pragma unroll
                for (int i = 0; i < 16; ++i) { xx[i] = (xx[i] - mean) * rstd; t1[i] = da[i] * xx[i]; }
                ac[6 * WV_D + lane] = wv_colsum(t1, c);
                ac[7 * WV_D + lane] = wv_colsum(da, c);
                wv_par_t(par + 0 * WV_D, g, pv);
                wv_ln_bwd(da, xx, pv, rstd, t1);
#pragma unroll
                for (int i = 0; i < 16; ++i) dx[i] = dx1[i] + t1[i];
            }
        }
        WV_MARK();
        // ---- embedding backward (re_sasrec_embed_bwd fused in): pad rows -> 0, the embedding's dropout mask, * sqrt(D)
        {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = dead ? 0.f : dx[i];
                if (thresh && !dead) v = ((emask >> i) & 1u) ? v * drop_scale : 0.f;
                dx[i] = v * emb_scale;
            }
            if (gid >= 0) wv_st_row(dOut + (int64_t)gid * WV_D, g, dx);
            wv_st_t(H.g_rows + row0 * WV_D, c, g, dx);
        }
        }   // active
        // ---- the item's vector gradients: the four waves' stages in wave order -> the workgroup's slab; the item's loss -> the ticket
        head_loss = re_wave_sum(head_loss);
        if (lane == 0) s_red[wave] = head_loss;
        __syncthreads();
        {
            float* sl = slab + (size_t)blockIdx.x * L * EG_NVEC * WV_D;
            for (int e = tid; e < L * EG_NVEC * WV_D; e += WV_NT) {
                const int f = e % WV_D, src = (e - f) + wv_colsum_lane(f);
                const float s = ((s_acc[src] + s_acc[(size_t)L * EG_NVEC * WV_D + src]) + s_acc[(size_t)2 * L * EG_NVEC * WV_D + src]) +
                                s_acc[(size_t)3 * L * EG_NVEC * WV_D + src];
                sl[e] = (k == 0) ? s : sl[e] + s;
            }
        }
        if (tid == 0) {
            const double part = (double)(((s_red[0] + s_red[1]) + s_red[2]) + s_red[3]);
            const bool finite = part == part && fabs(part) < 4294967296.0;
            const unsigned long long add = finite ? (unsigned long long)(long long)llrint(part * 1073741824.0) : 0ull;
            const unsigned long long old = __hip_atomic_fetch_add(H.acc, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long one = 1ull + (finite ? 0ull : (1ull << 32)) + (old & 0ull);
            const unsigned long long ticket = __hip_atomic_fetch_add(H.acc + 1, one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((int)(ticket & 0xFFFFFFFFull) == n_items - 1) {
                const unsigned long long tot = __hip_atomic_exchange(H.acc, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const bool bad = ((ticket + one) >> 32) != 0ull;
                const int cnt = H.count[0];
                H.loss[0] = (cnt > 0 && !bad) ? (float)((double)(long long)tot * (1.0 / 1073741824.0) / (double)cnt) : __builtin_nanf("");
                __hip_atomic_store(H.acc + 1, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();   // (the stage and the loss partials are free for the next item)
    }
}

int enc_wave_step_launch(const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P, float ds, uint32_t thresh,
                         uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan, int grid, const EncHead& H, float* dx0,
                         float* gtape, float* slab, float scale, uint32_t* wf, hipStream_t s) {
    const EncTape T = enc_tape_layout(B, S, WV_D, L);
    hipLaunchKernelGGL(enc_wave_prep_k, dim3((unsigned)(6 * L * 1024 / 256)), dim3(256), 0, s, P, (int)L, wf);
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    const size_t ldsb = wv_lds_floats((int)L) * sizeof(float);
    if (hipFuncSetAttribute((const void*)enc_wave_step_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
    hipLaunchKernelGGL(enc_wave_step_k, dim3(grid), dim3(WV_NT), ldsb, s, em, seq, (int)B, (int)S, (int)L, P, ds, thresh, seed, u, (float*)tape, T, plan, H,
                       dx0, gtape, slab, seed_dev, scale, (const uint32_t*)wf);
    return hipGetLastError() == hipSuccess ? RE_OK : RE_ELAUNCH;
}
