// General fp32 GEMM on the matrix cores:  C = alpha * op(A) op(B) + beta * C (+ bias[n]) (+ ReLU),  op = identity / transpose.
//
// Used where the reference calls nn.Linear / einsum with shapes that are not per-sequence tiles:
//   * SASRec --loss CE: logits = u E^T [M, N] and the two backward products (SASRec/main.py:217-219),
//   * DeepFM's MLP 100 -> 400 -> 400 -> 400 -> 1 (DeepFM/main.py:103-124,151-164), forward and backward.
// v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain per output).  One workgroup = 64 x 64 outputs, 4 waves each
// owning a 64 x 16 column strip; K is consumed in steps of 32 through LDS.  Both operands are staged k-contiguous
// (As[m][k], Bs[n][k], 16-B row pad), so every fragment is two ds_read_b128 whatever the memory layout: the staging
// pass does the transposition (float4 global loads along the operand's contiguous dimension).  Skinny outputs with a
// long K (dU = dlogits E: 3 000 x 64 x 12 101) are split along K over gridDim.z into partial slabs that a second
// kernel adds in slice order (deterministic; no float atomics).
#include <type_traits>

#include "re_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define GM_BM 64
#define GM_BN 64
#define GM_BK 32
#define GM_LS (GM_BK + 4)

// stage a [64 x 32] tile of op(X) into Xs[r][k]: X(r, k) = trans ? mem[k * ld + r] : mem[r * ld + k]
__device__ __forceinline__ void gm_stage(float* Xs, const float* __restrict__ mem, int64_t ld, bool trans, int64_t r0, int64_t k0,
                                         int64_t R, int64_t K, int tid, bool vec_ok) {
    if (!trans) {  // k contiguous in memory: thread -> (row, 4 consecutive k)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = p * 256 + tid;           // 512 float4 slots: 64 rows x 8
            const int r = f >> 3, kq = (f & 7) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t gr = r0 + r, gk = k0 + kq;
            if (gr < R) {
                const float* src = mem + gr * ld + gk;
                if (vec_ok && gk + 3 < K) v = *reinterpret_cast<const float4*>(src);
                else {
                    if (gk + 0 < K) v.x = src[0];
                    if (gk + 1 < K) v.y = src[1];
                    if (gk + 2 < K) v.z = src[2];
                    if (gk + 3 < K) v.w = src[3];
                }
            }
            *reinterpret_cast<float4*>(Xs + r * GM_LS + kq) = v;
        }
    } else {      // r contiguous in memory: thread -> (k, 4 consecutive rows)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int f = p * 256 + tid;           // 512 float4 slots: 32 k x 16
            const int k = f >> 4, rq = (f & 15) * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            const int64_t gk = k0 + k, gr = r0 + rq;
            if (gk < K) {
                const float* src = mem + gk * ld + gr;
                if (vec_ok && gr + 3 < R) v = *reinterpret_cast<const float4*>(src);
                else {
                    if (gr + 0 < R) v.x = src[0];
                    if (gr + 1 < R) v.y = src[1];
                    if (gr + 2 < R) v.z = src[2];
                    if (gr + 3 < R) v.w = src[3];
                }
            }
            Xs[(rq + 0) * GM_LS + k] = v.x;
            Xs[(rq + 1) * GM_LS + k] = v.y;
            Xs[(rq + 2) * GM_LS + k] = v.z;
            Xs[(rq + 3) * GM_LS + k] = v.w;
        }
    }
}

// The same tile in two halves -- global -> registers (unconditional 16-byte loads: only for a tile that lies wholly inside the operand) and
// registers -> LDS -- so that the NEXT k step's tile is in flight while this step's products run: staged in one go, every k step exposed a
// memory round trip (DeepFM's 4096 x 400 x 400 products: 13 steps, 28 us at 0.3 of the fp32 matrix peak).
__device__ __forceinline__ void gm_fetch(float4 (&v)[2], const float* __restrict__ mem, int64_t ld, bool trans, int64_t r0, int64_t k0, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int f = p * 256 + tid;
        v[p] = trans ? *reinterpret_cast<const float4*>(mem + (k0 + (f >> 4)) * ld + r0 + (f & 15) * 4)
                     : *reinterpret_cast<const float4*>(mem + (r0 + (f >> 3)) * ld + k0 + (f & 7) * 4);
    }
}
__device__ __forceinline__ void gm_put(float* Xs, const float4 (&v)[2], bool trans, int tid) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int f = p * 256 + tid;
        if (!trans) *reinterpret_cast<float4*>(Xs + (f >> 3) * GM_LS + (f & 7) * 4) = v[p];
        else {
            const int k = f >> 4, rq = (f & 15) * 4;
            Xs[(rq + 0) * GM_LS + k] = v[p].x;
            Xs[(rq + 1) * GM_LS + k] = v[p].y;
            Xs[(rq + 2) * GM_LS + k] = v[p].z;
            Xs[(rq + 3) * GM_LS + k] = v[p].w;
        }
    }
}

__global__ __launch_bounds__(256) void gemm_f32_k(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha,
                                                  const float* __restrict__ A, int64_t lda, const float* __restrict__ B, int64_t ldb,
                                                  float beta, float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                                  int relu, float* __restrict__ slabs, int64_t kchunk, int vecA, int vecB) {
    __shared__ __align__(16) float As[GM_BM * GM_LS];
    __shared__ __align__(16) float Bs[GM_BN * GM_LS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int64_t m0 = (int64_t)blockIdx.x * GM_BM, n0 = (int64_t)blockIdx.y * GM_BN;
    const int64_t kb = (int64_t)blockIdx.z * kchunk;
    const int64_t ke = (kb + kchunk < K) ? kb + kchunk : K;
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (a tile can be requested ahead if it lies wholly inside its operand: rows m0 .. m0 + 63 / n0 .. n0 + 63 and a full k step)
    const bool inA = vecA != 0 && m0 + GM_BM <= M, inB = vecB != 0 && n0 + GM_BN <= N;
    float4 pa[2], pb[2];
    bool pre = false;                                   // (uniform) the registers hold this step's tiles
    for (int64_t k0 = kb; k0 < ke; k0 += GM_BK) {
        __syncthreads();
        if (pre) {
            gm_put(As, pa, transA != 0, tid);
            gm_put(Bs, pb, transB == 0, tid);
        } else {
            gm_stage(As, A, lda, transA != 0, m0, k0, M, ke, tid, vecA != 0);
            gm_stage(Bs, B, ldb, transB == 0, n0, k0, N, ke, tid, vecB != 0);   // op(B)(k, n): transB == 0 means n contiguous
        }
        __syncthreads();
        pre = inA && inB && k0 + 2 * GM_BK <= ke;       // the next step is a full one
        if (pre) {
            gm_fetch(pa, A, lda, transA != 0, m0, k0 + GM_BK, tid);
            gm_fetch(pb, B, ldb, transB == 0, n0, k0 + GM_BK, tid);
        }
        // lane group g owns k = 8g .. 8g+7 of this step (the same assignment for A and B)
        float bf[8];
        {
            const float* p = Bs + (16 * wave + c) * GM_LS + 8 * g;
            const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
            bf[0] = v0.x; bf[1] = v0.y; bf[2] = v0.z; bf[3] = v0.w; bf[4] = v1.x; bf[5] = v1.y; bf[6] = v1.z; bf[7] = v1.w;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float* p = As + (16 * t + c) * GM_LS + 8 * g;
            const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
            const float af[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
            for (int s = 0; s < 8; ++s) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[s], bf[s], acc[t], 0, 0, 0);
        }
    }
    const int64_t n = n0 + 16 * wave + c;
    if (n >= N) return;
    const float bv = (bias && !slabs) ? bias[n] : 0.f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + 16 * t + 4 * g + j;
            if (m >= M) continue;
            if (slabs) {
                slabs[((int64_t)blockIdx.z * M + m) * N + n] = acc[t][j];
            } else {
                float v = alpha * acc[t][j] + bv;
                if (beta != 0.f) v = fmaf(beta, C[m * ldc + n], v);
                if (relu) v = fmaxf(v, 0.f);
                C[m * ldc + n] = v;
            }
        }
}

__global__ __launch_bounds__(256) void gemm_splitk_reduce(const float* __restrict__ slabs, int nsplit, int64_t M, int64_t N, float alpha,
                                                          float beta, float* __restrict__ C, int64_t ldc, const float* __restrict__ bias,
                                                          int relu) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= M * N) return;
    const int64_t m = e / N, n = e - m * N;
    float s = 0.f;
    for (int z = 0; z < nsplit; ++z) s += slabs[(int64_t)z * M * N + e];
    float v = alpha * s + (bias ? bias[n] : 0.f);
    if (beta != 0.f) v = fmaf(beta, C[m * ldc + n], v);
    if (relu) v = fmaxf(v, 0.f);
    C[m * ldc + n] = v;
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// The WIDE form (DeepFM's MLP shapes: 4096 x 400 x {100, 400} forward / input gradient, 400 x {100, 400} x 4096 weight gradient).
//
// What the 64 x 64 form above leaves on the table at these shapes (0.32 of the fp32 matrix peak, round 4): N = 400 is 6.25 column blocks of
// 64 (448 workgroups on 256 CUs: two rounds, the second a quarter full), a wave's 8 MFMAs of a tile run back to back on ONE accumulator
// (40 dependent cycles each instead of 32 issue cycles), two barriers per k step, and operands that are not k-contiguous are transposed
// on their way into LDS with 8-way conflicted scalar stores.
//
// Here a workgroup owns BM = 16 WM rows and up to TT column tiles of 16 (WM = 4: wave w = row tile w, all TT column tiles, TT accumulators
// -- the k sub-steps visit them round robin, so no MFMA waits for the one before it; WM = 1: 16 rows, the four waves take two column
// tiles each: the form for N <= 128, where 64-row blocks would leave most CUs without a workgroup).  The N / 16 column tiles are dealt over
// the column blocks as evenly as they go (N = 400: 25 tiles = 7 + 6 + 6 + 6 -> 64 x 4 = 256 workgroups, one per CU, 7 tile-times against
// 6.25 ideal).  K runs in chunks of 64 through DOUBLE-BUFFERED LDS: the next chunk's global loads are issued before the chunk's MFMAs, land
// in registers meanwhile, are written to the other buffer behind them; one barrier per chunk.  Inside a chunk the fragments of sub-chunk
// q + 1 are read from LDS into a second register set BEFORE sub-chunk q's MFMAs are issued (one wave per SIMD: nothing else hides an LDS
// round trip), and sub-chunks beyond K are skipped (K = 400 = 6 chunks + 1 sub-chunk: no padded work).
// An operand is staged in the orientation it has in memory (16-byte loads and 16-byte LDS stores either way):
//   k-contiguous  -> Xs[row][36]:      a lane's fragment of a 16-k sub-chunk is ONE ds_read_b128 (k = 16 q + 4 g + i for MFMA i of lane group g)
//   row-contiguous -> Xs[k][rows + 4]: four ds_read_b32 at rows k = 16 q + 4 g + i (row stride = 4 mod 8 floats: lane groups g land on
//                                      different bank quarters, conflict-free)
// -- the same k(g, i) in both, so any pairing of the two orientations multiplies matching k.  Exact fp32 (v_mfma_f32_16x16x4_f32); the
// order of a dot product's terms is a permutation of the 64 x 64 form's (both fixed: deterministic).
// Epilogue: alpha / beta / bias / ReLU as above, split-K slabs, and (WM = 4, M a multiple of 64, no split) the BatchNorm batch statistics of
// the output's columns as per-64-row (mean, M2) partials in re_bn_relu_drop_fwd's workspace format -- bn_stats_partial_k's pass over z
// (DeepFM/main.py:119-124: bn(linear(x))) is the GEMM's own epilogue.
#define GW_BK 64
// scheduling directions for one stretch of a basic block: NOPS instructions of class MASK (0x100 LDS read, 0x200 LDS write) spread evenly over
// NMF MFMAs (0x008), in program order of each class
template <int MASK, int NOPS, int NMF>
__device__ __forceinline__ void gw_spread() {
    if constexpr (NOPS <= NMF) {
        constexpr int per = NMF / NOPS;
#pragma unroll
        for (int i = 0; i < NOPS; ++i) {
            __builtin_amdgcn_sched_group_barrier(MASK, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, per, 0);
        }
        if constexpr (NMF - NOPS * per > 0) __builtin_amdgcn_sched_group_barrier(0x008, NMF - NOPS * per, 0);
    } else {
        constexpr int per = (NOPS + NMF - 1) / NMF;
#pragma unroll
        for (int i = 0; i < NMF; ++i) {
            __builtin_amdgcn_sched_group_barrier(MASK, per, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
    }
}
#define GW_LSK (GW_BK + 4)          // row stride of a k-contiguous tile
template <bool KMAJ, int ROWS>      // ROWS: rows (or columns) of the tile: a multiple of 16
struct GwTile {
    static constexpr int LSR = ROWS + 4;                                  // row stride of a row-contiguous tile ([k][rows])
    static constexpr int FLOATS = KMAJ ? ROWS * GW_LSK : GW_BK * LSR;
    static constexpr int SLOTS = ROWS * GW_BK / 4;                        // float4 slots of a chunk
    static constexpr int PER = (SLOTS + 255) / 256;                       // ... per thread
    // global -> registers.  mem: the operand's base; (r0, k0): the tile's first row / k; rows [r0, r_end) and k [k0, k_end) exist.
    // contiguous dimension in multiples of 4 and 16-byte aligned (the launcher checks): a float4 is inside or outside as a whole.
    static __device__ __forceinline__ void fetch(float4 (&v)[PER], const float* __restrict__ mem, int64_t ld, int64_t r0, int64_t r_end,
                                                 int64_t k0, int64_t k_end, int tid, bool interior) {
        if (interior) {            // (uniform: the tile lies wholly inside the operand -- unconditional loads, no per-load branch)
#pragma unroll
            for (int p = 0; p < PER; ++p) {
                const int f = p * 256 + tid;
                if (SLOTS % 256 != 0 && f >= SLOTS) { v[p] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
                if (KMAJ) v[p] = *reinterpret_cast<const float4*>(mem + (r0 + (f / (GW_BK / 4))) * ld + k0 + (f % (GW_BK / 4)) * 4);
                else v[p] = *reinterpret_cast<const float4*>(mem + (k0 + f / (ROWS / 4)) * ld + r0 + (f % (ROWS / 4)) * 4);
            }
            return;
        }
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = p * 256 + tid;
            v[p] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (SLOTS % 256 != 0 && f >= SLOTS) continue;
            if (KMAJ) {
                const int64_t r = r0 + (f / (GW_BK / 4)), k = k0 + (f % (GW_BK / 4)) * 4;
                if (r < r_end && k < k_end) v[p] = *reinterpret_cast<const float4*>(mem + r * ld + k);
            } else {
                const int64_t k = k0 + f / (ROWS / 4), r = r0 + (f % (ROWS / 4)) * 4;
                if (r < r_end && k < k_end) v[p] = *reinterpret_cast<const float4*>(mem + k * ld + r);
            }
        }
    }
    // the slots' source pointers at k = k0 with the row clamped into [r0, r_end) (k-contiguous: the last row; row-contiguous: the last whole
    // float4 of rows) -- for steps that lie wholly inside the operand in k: fetch_at(v, gp, off) loads every slot at gp + off, no guards
    static __device__ __forceinline__ void ptrs(const float* (&gp)[PER], const float* __restrict__ mem, int64_t ld, int64_t r0, int64_t r_end,
                                                int64_t k0, int tid) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            int f = p * 256 + tid;
            if (SLOTS % 256 != 0 && f >= SLOTS) f = SLOTS - 1;
            if (KMAJ) {
                int64_t r = r0 + (f / (GW_BK / 4));
                if (r > r_end - 1) r = r_end - 1;
                gp[p] = mem + r * ld + k0 + (f % (GW_BK / 4)) * 4;
            } else {
                int64_t r = r0 + (f % (ROWS / 4)) * 4;
                if (r > r_end - 4) r = r_end - 4;
                gp[p] = mem + (k0 + f / (ROWS / 4)) * ld + r;
            }
        }
    }
    static __device__ __forceinline__ void fetch_at(float4 (&v)[PER], const float* const (&gp)[PER], int64_t off) {
#pragma unroll
        for (int p = 0; p < PER; ++p) v[p] = *reinterpret_cast<const float4*>(gp[p] + off);
    }
    static __device__ __forceinline__ void put(float* Xs, const float4 (&v)[PER], int tid) {
#pragma unroll
        for (int p = 0; p < PER; ++p) {
            const int f = p * 256 + tid;
            if (SLOTS % 256 != 0 && f >= SLOTS) continue;
            if (KMAJ) *reinterpret_cast<float4*>(Xs + (f / (GW_BK / 4)) * GW_LSK + (f % (GW_BK / 4)) * 4) = v[p];
            else *reinterpret_cast<float4*>(Xs + (f / (ROWS / 4)) * LSR + (f % (ROWS / 4)) * 4) = v[p];
        }
    }
    // the lane's fragment of 16-row tile `t`, sub-chunk q: element i = X(16 t + c, 16 q + 4 g + i)
    static __device__ __forceinline__ f32x4 frag(const float* Xs, int t, int q, int c, int g) {
        if (KMAJ) return *reinterpret_cast<const f32x4*>(Xs + (16 * t + c) * GW_LSK + 16 * q + 4 * g);
        const float* p = Xs + (16 * q + 4 * g) * LSR + 16 * t + c;
        return (f32x4){p[0], p[LSR], p[2 * LSR], p[3 * LSR]};
    }
};

struct GwArgs {
    int64_t M, N, K;
    float alpha, beta;
    const float *A, *B;
    int64_t lda, ldb, ldc;
    float* C;
    const float* bias;
    int relu;
    float* slabs;          // split-K: [gridDim.z][M][N] partial products (alpha / beta / bias / relu applied by gemm_splitk_reduce)
    int64_t kchunk;
    int nt_base, nt_rem;   // column tiles per column block: base (+ 1 for the first nt_rem blocks)
    float* colstats;       // [M / 64][2][N]: (mean, M2) of the output's columns over the block's 64 rows, or null
    // gated output (the backward of dropout(relu(bn(.))) in the epilogue of the product that makes its incoming gradient): with v the product,
    // C = g = act > 0 ? gate_scale v : 0 and colstats = the block's (sum g, sum g xhat), xhat = (z - mean) rstd from gate_stats [2][N]
    const float *gate_act, *gate_z, *gate_stats;
    float gate_scale;
};

template <bool A_KMAJ, bool B_KMAJ, int WM, int TT>
__global__ __launch_bounds__(256) void gemm_wide_k(const GwArgs a) {
    constexpr int BM = 16 * WM;
    constexpr int WT = (WM == 4) ? TT : 2;           // column tiles a wave multiplies
    constexpr int BT = (WM == 4) ? TT : 8;           // column tiles of the workgroup's B tile
    using TA = GwTile<A_KMAJ, BM>;
    using TB = GwTile<B_KMAJ, 16 * BT>;
    extern __shared__ __align__(16) float gw_lds[];
    float* const As0 = gw_lds;
    float* const Bs0 = gw_lds + 2 * TA::FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int cb = blockIdx.y;
    const int t_first = cb * a.nt_base + (cb < a.nt_rem ? cb : a.nt_rem), t_cnt = a.nt_base + (cb < a.nt_rem ? 1 : 0);
    const int64_t n0 = (int64_t)t_first * 16;
    const int64_t n_end = (n0 + 16 * t_cnt < a.N) ? n0 + 16 * t_cnt : a.N;
    const int64_t kb = (int64_t)blockIdx.z * a.kchunk;
    const int64_t ke = (kb + a.kchunk < a.K) ? kb + a.kchunk : a.K;
    const int arow = (WM == 4) ? wave : 0;           // the wave's row tile
    const int bcol0 = (WM == 4) ? 0 : 2 * wave;      // ... and its first column tile inside the block
    const bool in_a = m0 + BM <= a.M, in_b = n0 + 16 * BT <= n_end;     // (uniform) the row / column extent of the tiles is inside the operands
    f32x4 acc[WT];
#pragma unroll
    for (int t = 0; t < WT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 pa[TA::PER], pb[TB::PER];
    // ---- the k loop, software-pipelined for ONE wave per SIMD (93 KB of LDS a workgroup: no second wave hides a barrier or an LDS round trip).
    // Step s works on LDS buffer s & 1.  Between barrier(s - 1) and barrier(s) a wave issues, in this order,
    //     global loads of tile s + 2 -> registers;  fragments q0(s) <- LDS;  MFMAs q3(s - 1)            (the MFMAs cover both latencies)
    //     MFMAs q0(s) | fragments q1(s);  MFMAs q1(s) | fragments q2(s);  MFMAs q2(s) | fragments q3(s) + tile s + 1: registers -> LDS
    // so the only thing an MFMA ever waits for is the barrier itself.  (The plain loop -- fetch, four sub-chunks, store, barrier -- measured
    // 54 % MFMA-busy at K = 4 096: 1 300 cycles a step parked at waits, 1 700 issuing everything else with the MFMA pipe idle.)
    // Steps 0 .. last - 1 are whole (GW_BK of k); the last may be partial and runs guarded.  Rows / columns outside the operand are CLAMPED into
    // it for the whole steps (unconditional loads; what a clamped slot holds lands in accumulator rows / columns the epilogue never stores);
    // the partial step's loads are guarded and zero-filled.
    const int64_t klen = ke - kb;
    const int nfull = (int)(klen / GW_BK);
    const int nsteps = nfull + ((klen % GW_BK) ? 1 : 0);
    const int last = nsteps - 1;
    const float *gpa[TA::PER], *gpb[TB::PER];
    TA::ptrs(gpa, a.A, a.lda, m0, a.M, kb, tid);
    TB::ptrs(gpb, a.B, a.ldb, n0, n_end, kb, tid);
    const int64_t ska = A_KMAJ ? (int64_t)GW_BK : (int64_t)GW_BK * a.lda, skb = B_KMAJ ? (int64_t)GW_BK : (int64_t)GW_BK * a.ldb;   // a step, in floats
    auto fetch = [&](int s_) {                        // tile s_ -> registers (s_ <= last; uniform)
        if (s_ < nfull) {
            TA::fetch_at(pa, gpa, (int64_t)s_ * ska);
            TB::fetch_at(pb, gpb, (int64_t)s_ * skb);
        } else {
            TA::fetch(pa, a.A, a.lda, m0, a.M, kb + (int64_t)s_ * GW_BK, ke, tid, false);
            TB::fetch(pb, a.B, a.ldb, n0, n_end, kb + (int64_t)s_ * GW_BK, ke, tid, false);
        }
    };
    f32x4 av[2], bv[2][WT];
    auto frags = [&](int buf, int q, int slot) {
        av[slot] = TA::frag(As0 + buf * TA::FLOATS, arow, q, c, g);
#pragma unroll
        for (int t = 0; t < WT; ++t) bv[slot][t] = TB::frag(Bs0 + buf * TB::FLOATS, bcol0 + t, q, c, g);
    };
    auto mfmas = [&](int slot) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int t = 0; t < WT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[slot][i], bv[slot][t][i], acc[t], 0, 0, 0);
    };
    // LDS reads of a sub-chunk's fragments (k-contiguous: one ds_read_b128 each; else four b32), its MFMAs, the LDS writes of a tile
    constexpr int RDA = A_KMAJ ? 1 : 4, RDB = B_KMAJ ? 1 : 4, NRDI = RDA + WT * RDB, NMF = 4 * WT, NWR = TA::PER + TB::PER;
    fetch(0);
    TA::put(As0, pa, tid);
    TB::put(Bs0, pb, tid);
    __syncthreads();
    if (nsteps > 1) fetch(1);
    frags(0, 0, 0);
    // a whole step with a successor.  FAST: tile s_ + 2 exists and is whole (its loads unconditional, no branch in the step); otherwise it is
    // the partial last tile (guarded loads) or there is none -- the two steps in front of the last one
    auto step = [&](int s_, auto fast_tag) {
        constexpr bool FAST = decltype(fast_tag)::value;
        const int buf = s_ & 1;
        frags(buf, 1, 1);
        mfmas(0);
        frags(buf, 2, 0);
        mfmas(1);
        frags(buf, 3, 1);
        TA::put(As0 + (buf ^ 1) * TA::FLOATS, pa, tid);
        TB::put(Bs0 + (buf ^ 1) * TB::FLOATS, pb, tid);
        mfmas(0);
        // the order of issue: a sub-chunk's LDS reads spread over the MFMAs of the one before it, the tile's LDS writes over the third's
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        gw_spread<0x100, NRDI, NMF - 1>();
        gw_spread<0x100, NRDI, NMF>();
        gw_spread<0x100, NRDI, NMF - NMF / 2>();
        gw_spread<0x200, NWR, NMF / 2>();
        __syncthreads();
        if (FAST) {
            TA::fetch_at(pa, gpa, (int64_t)(s_ + 2) * ska);
            TB::fetch_at(pb, gpb, (int64_t)(s_ + 2) * skb);
        } else if (s_ + 2 <= last) {
            fetch(s_ + 2);
        }
        frags(buf ^ 1, 0, 0);
        mfmas(1);
        // (two wait states in front of the back edge: where the compiler rotates accumulators at the loop header -- the 64 x 64 form -- its copies
        //  read the last MFMA's result one wait state earlier than scripts/lint_mfma_hazard.py's table allows)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");
    };
    int s_ = 0;
    for (; s_ + 2 < nfull && s_ < last; ++s_) step(s_, std::true_type{});
    for (; s_ < last; ++s_) step(s_, std::false_type{});
    {                                                 // the last step: whole or partial, no successor (fragments q0 are in slot 0)
        const int buf = last & 1;
        const int64_t kl = ke - (kb + (int64_t)last * GW_BK);
        const int nq = (int)((kl + 15) >> 4) < GW_BK / 16 ? (int)((kl + 15) >> 4) : GW_BK / 16;
#pragma unroll
        for (int q = 0; q < GW_BK / 16; ++q) {
            if (q >= nq) break;                      // (uniform)
            if (q + 1 < GW_BK / 16 && q + 1 < nq) frags(buf, q + 1, (q + 1) & 1);
            mfmas(q & 1);
        }
        __syncthreads();                             // (the epilogue reuses the tile memory)
    }
    (void)in_a; (void)in_b;
    // ---- epilogue: acc[t][j] = C(m0 + 16 arow + 4 g + j, n0 + 16 (bcol0 + t) + c)
    float* stat = Bs0;                                              // [4 waves][16 BT columns][2] (the main loop is behind a barrier)
    if (WM == 4 && a.gate_act) {
        // (M a multiple of 64, no split: the launcher's conditions)
#pragma unroll
        for (int t = 0; t < WT; ++t) {
            const int64_t n = n0 + 16 * t + c;
            const bool n_ok = t < t_cnt && n < a.N;
            const int64_t nn = n_ok ? n : 0;
            const float mu = a.gate_stats[nn], rs = a.gate_stats[a.N + nn];
            float av[4], zv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int64_t e = (m0 + 16 * arow + 4 * g + j) * a.ldc + nn;
                av[j] = a.gate_act[e]; zv[j] = a.gate_z[e];
            }
            float sg = 0.f, sx = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gv = (n_ok && av[j] > 0.f) ? a.alpha * acc[t][j] * a.gate_scale : 0.f;
                if (n_ok) a.C[(m0 + 16 * arow + 4 * g + j) * a.ldc + n] = gv;
                sg += gv;
                sx = fmaf(gv, (zv[j] - mu) * rs, sx);
            }
            {
                auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(sg), __float_as_uint(sg), false, false);
                sg = __uint_as_float(x[0]) + __uint_as_float(x[1]);
                x = __builtin_amdgcn_permlane32_swap(__float_as_uint(sg), __float_as_uint(sg), false, false);
                sg = __uint_as_float(x[0]) + __uint_as_float(x[1]);
                x = __builtin_amdgcn_permlane16_swap(__float_as_uint(sx), __float_as_uint(sx), false, false);
                sx = __uint_as_float(x[0]) + __uint_as_float(x[1]);
                x = __builtin_amdgcn_permlane32_swap(__float_as_uint(sx), __float_as_uint(sx), false, false);
                sx = __uint_as_float(x[0]) + __uint_as_float(x[1]);
            }
            if (g == 0) *reinterpret_cast<float2*>(stat + ((wave * BT + t) * 16 + c) * 2) = make_float2(sg, sx);
        }
        __syncthreads();
        if (tid < 16 * BT) {
            const int64_t n = n0 + tid;
            if (tid < 16 * t_cnt && n < a.N) {
                float2 p[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) p[w] = *reinterpret_cast<const float2*>(stat + (w * BT * 16 + tid) * 2);
                a.colstats[((int64_t)blockIdx.x * 2 + 0) * a.N + n] = (p[0].x + p[1].x) + (p[2].x + p[3].x);
                a.colstats[((int64_t)blockIdx.x * 2 + 1) * a.N + n] = (p[0].y + p[1].y) + (p[2].y + p[3].y);
            }
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < WT; ++t) {
        const int64_t n = n0 + 16 * (bcol0 + t) + c;
        const bool n_ok = (bcol0 + t) < t_cnt && n < a.N;
        const float bv = (a.bias && !a.slabs && n_ok) ? a.bias[n] : 0.f;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t m = m0 + 16 * arow + 4 * g + j;
            v[j] = acc[t][j];
            if (!n_ok || m >= a.M) { v[j] = 0.f; continue; }
            if (a.slabs) { a.slabs[((int64_t)blockIdx.z * a.M + m) * a.N + n] = v[j]; continue; }
            v[j] = a.alpha * v[j] + bv;
            if (a.beta != 0.f) v[j] = fmaf(a.beta, a.C[m * a.ldc + n], v[j]);
            if (a.relu) v[j] = fmaxf(v[j], 0.f);
            a.C[m * a.ldc + n] = v[j];
        }
        if (WM == 4 && a.colstats) {
            // the column's 16 rows of this wave: mean, then M2 about it (two passes over registers); lane groups g combined by row swaps
            float s = (v[0] + v[1]) + (v[2] + v[3]);
            {
                const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
                s = __uint_as_float(x[0]) + __uint_as_float(x[1]);
                const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(s), __float_as_uint(s), false, false);
                s = __uint_as_float(y[0]) + __uint_as_float(y[1]);
            }
            const float mu = s * (1.0f / 16);
            float q2 = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const float d = v[j] - mu; q2 = fmaf(d, d, q2); }
            {
                const auto x = __builtin_amdgcn_permlane16_swap(__float_as_uint(q2), __float_as_uint(q2), false, false);
                q2 = __uint_as_float(x[0]) + __uint_as_float(x[1]);
                const auto y = __builtin_amdgcn_permlane32_swap(__float_as_uint(q2), __float_as_uint(q2), false, false);
                q2 = __uint_as_float(y[0]) + __uint_as_float(y[1]);
            }
            if (g == 0) *reinterpret_cast<float2*>(stat + ((wave * BT + t) * 16 + c) * 2) = make_float2(mu, q2);
        }
    }
    if (WM == 4 && a.colstats) {
        __syncthreads();
        // Chan's merge of the four waves' (mean, M2, 16 rows), ((0, 1), (2, 3)): equal counts -> mean = average, M2 += delta^2 * n / 2
        if (tid < 16 * BT) {
            const int64_t n = n0 + tid;
            if (tid < 16 * t_cnt && n < a.N) {
                float2 p[4];
#pragma unroll
                for (int w = 0; w < 4; ++w) p[w] = *reinterpret_cast<const float2*>(stat + (w * BT * 16 + tid) * 2);
                const float d01 = p[1].x - p[0].x, d23 = p[3].x - p[2].x;
                const float m01 = p[0].x + 0.5f * d01, m23 = p[2].x + 0.5f * d23;
                const float q01 = (p[0].y + p[1].y) + d01 * d01 * 8.0f, q23 = (p[2].y + p[3].y) + d23 * d23 * 8.0f;
                const float d = m23 - m01;
                a.colstats[((int64_t)blockIdx.x * 2 + 0) * a.N + n] = m01 + 0.5f * d;
                a.colstats[((int64_t)blockIdx.x * 2 + 1) * a.N + n] = (q01 + q23) + d * d * 16.0f;
            }
        }
    }
}

// Which form runs a product: the wide one when its 16-byte loads are legal and its grid fills the chip at least as well.
struct GwPlan { int use, wm, tt, cb, nt_base, nt_rem, nsplit; };
#ifndef GW_TT4_SLOTS
#define GW_TT4_SLOTS 512
#endif
static GwPlan gw_plan(int transA, int transB, int64_t M, int64_t N, int64_t K, const float* A, int64_t lda, const float* B, int64_t ldb,
                      bool want_stats) {
    GwPlan p{0, 4, 8, 1, 0, 0, 1};
    const bool a_kmaj = transA == 0, b_kmaj = transB != 0;
    // contiguous dimension of each operand: whole float4s, 16-byte aligned
    const int64_t a_c = a_kmaj ? K : M, b_c = b_kmaj ? K : N;
    if ((reinterpret_cast<uintptr_t>(A) & 15u) || (reinterpret_cast<uintptr_t>(B) & 15u) || (lda & 3) || (ldb & 3) || (a_c & 3) || (b_c & 3)) return p;
    if (M < 64 || N < 16 || K < 64) return p;                         // (tiny products: the 64 x 64 form's guards cover them)
    const int64_t NT = re_cdiv(N, 16);
    double best = 1e300;
    const int tts[3] = {4, 7, 8};
    for (int form = 0; form < 4; ++form) {                            // WM = 4 with TT in {4, 7, 8}; WM = 1 (two tiles a wave)
        const int wm = form < 3 ? 4 : 1, tt = form < 3 ? tts[form] : 8, per_wave = form < 3 ? tt : 2;
        if (want_stats && (wm != 4 || (M & 63))) continue;
        const int64_t cb = re_cdiv(NT, tt), mb = re_cdiv(M, 16 * wm);
        int ns = 1;
        if (!want_stats && mb * cb < 192 && K >= 1024) {              // long K, few tiles: split K (slices of >= 8 chunks)
            int64_t s = 256 / (mb * cb), maxs = K / 256;
            if (s > maxs) s = maxs;
            if (s > 64) s = 64;
            ns = s < 1 ? 1 : (int)s;
        }
        // (the 64 x 64 tile form -- TT = 4: 140 - 168 registers, 64 KB of LDS -- fits two workgroups on a CU: 512 of them are one round)
        const double rounds = (double)re_cdiv(mb * cb * ns, (wm == 4 && tt == 4) ? GW_TT4_SLOTS : 256);
        const double cost = rounds * per_wave * (double)re_cdiv(re_cdiv(K, ns), GW_BK) + (ns > 1 ? 3.0 : 0.0) + (wm == 1 ? 0.5 : 0.0);
        if (cost < best) {
            best = cost;
            p = GwPlan{1, wm, tt, (int)cb, (int)(NT / cb), (int)(NT % cb), ns};
        }
    }
    return p;
}

template <bool AK, bool BK_, int WM, int TT>
static void gw_launch1(const GwArgs& a, dim3 grid, hipStream_t s) {
    constexpr size_t lds = 2 * (GwTile<AK, 16 * WM>::FLOATS + GwTile<BK_, 16 * (WM == 4 ? TT : 8)>::FLOATS) * sizeof(float);
    auto k = gemm_wide_k<AK, BK_, WM, TT>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
}
template <bool AK, bool BK_>
static void gw_launch(const GwPlan& p, const GwArgs& a, dim3 grid, hipStream_t s) {
    if (p.wm == 1) gw_launch1<AK, BK_, 1, 8>(a, grid, s);
    else if (p.tt == 4) gw_launch1<AK, BK_, 4, 4>(a, grid, s);
    else if (p.tt == 7) gw_launch1<AK, BK_, 4, 7>(a, grid, s);
    else gw_launch1<AK, BK_, 4, 8>(a, grid, s);
}

static int gm_nsplit(int64_t M, int64_t N, int64_t K) {
    const int64_t tiles = re_cdiv(M, GM_BM) * re_cdiv(N, GM_BN);
    if (tiles >= 256 || K < 1024) return 1;
    int64_t s = 512 / tiles;
    const int64_t maxs = K / 256;     // (slices of >= 8 K-steps; K / 1024 left the DeepFM weight-gradient shape 400 x 400 x 4096 at 196 workgroups: 80 -> 3x us)
    if (s > maxs) s = maxs;
    if (s > 64) s = 64;
    return s < 1 ? 1 : (int)s;
}

extern "C" size_t re_gemm_f32_workspace_bytes(int64_t M, int64_t N, int64_t K) {
    // (an upper bound over both forms: the wide form's split is at most 64 slices too, and is only known with the operands' alignment)
    int ns = gm_nsplit(M, N, K);
    if (K >= 1024 && ns < 64) {
        const int64_t s = K / 256;
        ns = (int)(s > 64 ? 64 : (s > ns ? s : ns));
    }
    return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 256;
}

// nsplit_out (optional): the split-K partial products are LEFT in ws -- [*nsplit_out][M][N], alpha / beta / bias / relu not applied -- for a
// later reduction of several products in one launch (re_gemm_splitk_reduce_many); *nsplit_out = 1: no split, C is written as usual.
static int gemm_run(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda, const float* B,
                    int64_t ldb, float beta, float* C, int64_t ldc, const float* bias, int relu, void* ws, size_t ws_bytes, float* colstats,
                    hipStream_t s, const float* gate_act = nullptr, const float* gate_z = nullptr, const float* gate_stats = nullptr,
                    float gate_scale = 1.f, int32_t* nsplit_out = nullptr) {
    if (nsplit_out) *nsplit_out = 1;
    const GwPlan wp = gw_plan(transA, transB, M, N, K, A, lda, B, ldb, colstats != nullptr);
    if (colstats && !wp.use) return RE_EUNSUPPORTED;
    if (wp.use) {
        if (wp.nsplit > 1 && (!ws || ws_bytes < (size_t)wp.nsplit * M * N * sizeof(float))) return RE_EWORKSPACE;
        int64_t kchunk = re_cdiv(re_cdiv(K, wp.nsplit), GW_BK) * GW_BK;
        if (kchunk < GW_BK) kchunk = GW_BK;
        const int ns = (int)re_cdiv(K, kchunk);
        GwArgs a{M, N, K, alpha, beta, A, B, lda, ldb, ldc, C, bias, relu, ns > 1 ? (float*)ws : nullptr, kchunk, wp.nt_base, wp.nt_rem, colstats,
                 gate_act, gate_z, gate_stats, gate_scale};
        dim3 grid((unsigned)re_cdiv(M, 16 * wp.wm), (unsigned)wp.cb, (unsigned)ns);
        if (transA == 0 && transB != 0) gw_launch<true, true>(wp, a, grid, s);
        else if (transA == 0) gw_launch<true, false>(wp, a, grid, s);
        else if (transB != 0) gw_launch<false, true>(wp, a, grid, s);
        else gw_launch<false, false>(wp, a, grid, s);
        if (ns > 1 && nsplit_out) { *nsplit_out = ns; return re_launch_status(); }
        if (ns > 1)
            hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)re_cdiv(M * N, 256)), dim3(256), 0, s, (const float*)ws, ns, M, N, alpha, beta,
                               C, ldc, bias, relu);
        return re_launch_status();
    }
    const int ns = gm_nsplit(M, N, K);
    if (ns > 1 && (!ws || ws_bytes < (size_t)ns * M * N * sizeof(float))) return RE_EWORKSPACE;
    int64_t kchunk = re_cdiv(re_cdiv(K, ns), GM_BK) * GM_BK;
    if (kchunk < GM_BK) kchunk = GM_BK;
    const int vecA = ((reinterpret_cast<uintptr_t>(A) & 15u) == 0 && (lda & 3) == 0) ? 1 : 0;
    const int vecB = ((reinterpret_cast<uintptr_t>(B) & 15u) == 0 && (ldb & 3) == 0) ? 1 : 0;
    dim3 grid((unsigned)re_cdiv(M, GM_BM), (unsigned)re_cdiv(N, GM_BN), (unsigned)ns);
    hipLaunchKernelGGL(gemm_f32_k, grid, dim3(256), 0, s, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, relu,
                       ns > 1 ? (float*)ws : (float*)nullptr, kchunk, vecA, vecB);
    if (ns > 1 && nsplit_out) { *nsplit_out = ns; return re_launch_status(); }
    if (ns > 1)
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)re_cdiv(M * N, 256)), dim3(256), 0, s, (const float*)ws, ns, M, N, alpha, beta,
                           C, ldc, bias, relu);
    return re_launch_status();
}

// Several split-K products' reductions as ONE launch (DeepFM's three weight-gradient products of a step: 400 x 400 x 4096 and the like are
// 16 slabs each; as launches of their own the reductions were three dispatches of ~5 us).  The sum of a product's slabs is the slab-order
// one of gemm_splitk_reduce: the same bits.
#define GR_MAX 8
struct GrMany {
    const float* slabs[GR_MAX];
    float* C[GR_MAX];
    long long MN[GR_MAX], N[GR_MAX], ldc[GR_MAX];
    float alpha[GR_MAX];
    int nsplit[GR_MAX];
    unsigned first[GR_MAX + 1];
    int n;
};
__global__ __launch_bounds__(256) void gemm_splitk_reduce_many_k(GrMany G) {
    int q = 0;
#pragma unroll
    for (int i = 1; i < GR_MAX; ++i) q += (i < G.n && blockIdx.x >= G.first[i]) ? 1 : 0;
    const long long e = (long long)(blockIdx.x - G.first[q]) * 256 + threadIdx.x;
    if (e >= G.MN[q]) return;
    float s = 0.f;
    for (int z = 0; z < G.nsplit[q]; ++z) s += G.slabs[q][(long long)z * G.MN[q] + e];
    const long long m = e / G.N[q], n = e - m * G.N[q];
    G.C[q][m * G.ldc[q] + n] = G.alpha[q] * s;
}
extern "C" int re_gemm_splitk_reduce_many(int32_t n, const float* const* slabs, const int32_t* nsplit, const int64_t* M, const int64_t* N,
                                          const float* alpha, float* const* C, const int64_t* ldc, re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (n < 0 || n > GR_MAX || !slabs || !nsplit || !M || !N || !alpha || !C || !ldc) return RE_EINVAL;
    GrMany G{};
    G.n = n;
    unsigned total = 0;
    for (int i = 0; i < n; ++i) {
        if (!slabs[i] || !C[i] || nsplit[i] < 1 || M[i] <= 0 || N[i] <= 0 || ldc[i] < N[i]) return RE_EINVAL;
        G.slabs[i] = slabs[i]; G.C[i] = C[i]; G.MN[i] = M[i] * N[i]; G.N[i] = N[i]; G.ldc[i] = ldc[i]; G.alpha[i] = alpha[i]; G.nsplit[i] = nsplit[i];
        G.first[i] = total;
        total += (unsigned)re_cdiv(M[i] * N[i], 256);
    }
    G.first[n] = total;
    hipLaunchKernelGGL(gemm_splitk_reduce_many_k, dim3(total), dim3(256), 0, (hipStream_t)stream, G);
    return re_launch_status();
}

// re_gemm_f32 (beta = 0, no bias / relu) whose split-K reduction is left to the caller: *nsplit_out > 1: the partial products are in ws
// ([nsplit][M][N], alpha not applied) and C is NOT written -- pass them to re_gemm_splitk_reduce_many; *nsplit_out == 1: C holds the product.
extern "C" int re_gemm_f32_slabs(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, float* C, int64_t ldc, void* ws, size_t ws_bytes, int32_t* nsplit_out,
                                 re_stream_t stream) {
    re_clear_error();
    if (!nsplit_out) return RE_EINVAL;
    *nsplit_out = 1;
    if (M == 0 || N == 0) return RE_OK;
    if (!A || !B || !C || M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) return RE_EINVAL;
    return gemm_run(transA, transB, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, nullptr, 0, ws, ws_bytes, nullptr, (hipStream_t)stream, nullptr, nullptr,
                    nullptr, 1.f, nsplit_out);
}

extern "C" int re_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                           const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* bias, int relu, void* ws,
                           size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (M == 0 || N == 0) return RE_OK;
    if (!A || !B || !C || M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) return RE_EINVAL;
    return gemm_run(transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, relu, ws, ws_bytes, nullptr, (hipStream_t)stream);
}

// re_gemm_f32 + the BatchNorm batch statistics of C's columns as per-64-row (mean, M2) partials: colstats [M / 64][2][N], the workspace
// format of re_bn_relu_drop_fwd (pass it as that call's `ws` with `stats_ready` = M / 64: its statistics pass over z is then skipped).
// M a multiple of 64, operands 16-byte aligned with leading dimensions in multiples of 4; otherwise RE_EUNSUPPORTED (the caller then runs
// re_gemm_f32 and lets re_bn_relu_drop_fwd take its own statistics).
extern "C" int re_gemm_f32_colstats(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                                    const float* B, int64_t ldb, float* C, int64_t ldc, const float* bias, float* colstats,
                                    re_stream_t stream) {
    re_clear_error();
    if (!A || !B || !C || !colstats || M <= 0 || N <= 0 || K <= 0 || lda < 1 || ldb < 1 || ldc < N) return RE_EINVAL;
    if (M & 63) return RE_EUNSUPPORTED;
    return gemm_run(transA, transB, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, bias, 0, nullptr, 0, colstats, (hipStream_t)stream);
}

// The product whose result is the gradient arriving at a dropout(relu(bn(z))) block, with that block's gate and its two column sums in the
// epilogue (DeepFM/main.py:119-124 backward): C = g = act > 0 ? drop_scale alpha op(A) op(B) : 0 (act: the block's OUTPUT -- positive exactly
// where relu passed and dropout kept), part [M / 64][2][N] = per-64-row (sum g, sum g xhat), xhat = (z - stats[0][n]) stats[1][n] -- what
// re_bn_bwd_apply takes.  act and z share C's leading dimension.  M a multiple of 64 and the wide form's alignment, else RE_EUNSUPPORTED.
extern "C" int re_gemm_f32_gated(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                                 const float* B, int64_t ldb, float* C, int64_t ldc, const float* act, const float* z, const float* stats,
                                 float drop_scale, float* part, re_stream_t stream) {
    re_clear_error();
    if (!A || !B || !C || !act || !z || !stats || !part || M <= 0 || N <= 0 || K <= 0 || lda < 1 || ldb < 1 || ldc < N) return RE_EINVAL;
    if (M & 63) return RE_EUNSUPPORTED;
    return gemm_run(transA, transB, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, nullptr, 0, nullptr, 0, part, (hipStream_t)stream, act, z, stats,
                    drop_scale);
}
