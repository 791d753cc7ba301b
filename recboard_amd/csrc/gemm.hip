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
    const int ns = gm_nsplit(M, N, K);
    return ns > 1 ? (size_t)ns * M * N * sizeof(float) : 256;
}

extern "C" int re_gemm_f32(int transA, int transB, int64_t M, int64_t N, int64_t K, float alpha, const float* A, int64_t lda,
                           const float* B, int64_t ldb, float beta, float* C, int64_t ldc, const float* bias, int relu, void* ws,
                           size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (M == 0 || N == 0) return RE_OK;
    if (!A || !B || !C || M < 0 || N < 0 || K < 0 || lda < 1 || ldb < 1 || ldc < N) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int ns = gm_nsplit(M, N, K);
    if (ns > 1 && (!ws || ws_bytes < re_gemm_f32_workspace_bytes(M, N, K))) return RE_EWORKSPACE;
    int64_t kchunk = re_cdiv(re_cdiv(K, ns), GM_BK) * GM_BK;
    if (kchunk < GM_BK) kchunk = GM_BK;
    const int vecA = ((reinterpret_cast<uintptr_t>(A) & 15u) == 0 && (lda & 3) == 0) ? 1 : 0;
    const int vecB = ((reinterpret_cast<uintptr_t>(B) & 15u) == 0 && (ldb & 3) == 0) ? 1 : 0;
    dim3 grid((unsigned)re_cdiv(M, GM_BM), (unsigned)re_cdiv(N, GM_BN), (unsigned)ns);
    hipLaunchKernelGGL(gemm_f32_k, grid, dim3(256), 0, s, transA, transB, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, relu,
                       ns > 1 ? (float*)ws : (float*)nullptr, kchunk, vecA, vecB);
    if (ns > 1)
        hipLaunchKernelGGL(gemm_splitk_reduce, dim3((unsigned)re_cdiv(M * N, 256)), dim3(256), 0, s, (const float*)ws, ns, M, N, alpha, beta,
                           C, ldc, bias, relu);
    return re_launch_status();
}
