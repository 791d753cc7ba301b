// K10: dense Adam with coupled L2 weight decay over a flat fp32 parameter arena.  HBM-bound:
// 16 B read + 12 B written per element.  torch.optim.Adam single-tensor formulation:
//   g += wd*p; m = b1*m + (1-b1)*g; v = b2*v + (1-b2)*g*g; p -= (lr/(1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
#include <math.h>

#include "re_common.h"

__global__ __launch_bounds__(256) void adam_vec4(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                 float4* __restrict__ v, int64_t n4, float b1, float b2, float omb1, float omb2,
                                                 float step_size, float inv_sqrt_bc2, float eps, float wd) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
#define RE_ADAM1(c) re_adam1(P.c, M.c, V.c, G.c, b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd);
        RE_ADAM1(x) RE_ADAM1(y) RE_ADAM1(z) RE_ADAM1(w)
#undef RE_ADAM1
        p[i] = P; m[i] = M; v[i] = V;
    }
}

__global__ __launch_bounds__(256) void adam_tail(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                 float* __restrict__ v, int64_t begin, int64_t n, float b1, float b2, float omb1,
                                                 float omb2, float step_size, float inv_sqrt_bc2, float eps, float wd) {
    int64_t i = begin + threadIdx.x;
    if (i >= n) return;
    float P = p[i], M = m[i], V = v[i];
    re_adam1(P, M, V, g[i], b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd);
    p[i] = P; m[i] = M; v[i] = V;
}

// hipGraph-friendly variant: the two step-dependent scalars come from device memory (hyper[0] = lr / (1 - beta1^t),
// hyper[1] = 1 / sqrt(1 - beta2^t)), so a captured step can be replayed with a fresh 8-byte upload per step.
__global__ __launch_bounds__(256) void adam_vec4_dev(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                                     float4* __restrict__ v, int64_t n4, float b1, float b2, float omb1, float omb2,
                                                     const float* __restrict__ hyper, float eps, float wd) {
    const float step_size = hyper[0], inv_sqrt_bc2 = hyper[1];
    if (inv_sqrt_bc2 == 0.f) return;   // {0, 0}: the caller gated this step off (parameters AND moments stay as they are)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
#define RE_ADAM1(c) re_adam1(P.c, M.c, V.c, G.c, b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd);
        RE_ADAM1(x) RE_ADAM1(y) RE_ADAM1(z) RE_ADAM1(w)
#undef RE_ADAM1
        p[i] = P; m[i] = M; v[i] = V;
    }
}

__global__ void step_state_k(uint32_t* __restrict__ state, uint32_t seed, float step_size, float inv_sqrt_bc2) {
    state[0] = seed; state[1] = 0u; state[2] = __float_as_uint(step_size); state[3] = __float_as_uint(inv_sqrt_bc2);
}

extern "C" int re_step_state(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, re_stream_t stream) {
    re_clear_error();
    if (!state || step < 1) return RE_EINVAL;
    hipLaunchKernelGGL(step_state_k, dim3(1), dim3(1), 0, (hipStream_t)stream, state, seed, (float)(lr / (1.0 - pow(beta1, (double)step))),
                       (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step))));
    return re_launch_status();
}

// ---- what stands in FRONT of a captured step, as ONE launch: the step's scalars (re_step_state) and the batch brought into the graph's static
// buffers -- byte copies, int64 -> fp32 casts (labels), zero fills (a gradient table the step's kernels only write where the batch points).
// As separate launches -- two or three copy_() calls, a cast, re_step_state, a fill -- they were 4 - 6 dispatches of ~5 us in front of every
// replay: 20 of DeepFM's 320 us, 15 of MF-BPR's 45.
#define ST_MAX 8
#define ST_CHUNK 4096            // bytes of destination a workgroup handles (256 threads x 16 B)
struct StageSegs {
    void* dst[ST_MAX];
    const void* src[ST_MAX];
    unsigned long long bytes[ST_MAX];      // of the DESTINATION
    int kind[ST_MAX];                      // 0 copy, 1 int64 -> fp32, 2 zero
    unsigned first[ST_MAX + 1];            // first workgroup of every segment
    int n;
};
__global__ __launch_bounds__(256) void step_stage_inputs_k(uint32_t* __restrict__ state, uint32_t seed, float step_size, float inv_sqrt_bc2,
                                                           StageSegs S) {
    if (blockIdx.x == 0 && threadIdx.x == 0 && state) {
        state[0] = seed; state[1] = 0u; state[2] = __float_as_uint(step_size); state[3] = __float_as_uint(inv_sqrt_bc2);
    }
    int sg = 0;
#pragma unroll
    for (int i = 1; i < ST_MAX; ++i) sg += (i < S.n && blockIdx.x >= S.first[i]) ? 1 : 0;
    if (sg >= S.n) return;
    const unsigned long long off = (unsigned long long)(blockIdx.x - S.first[sg]) * ST_CHUNK + threadIdx.x * 16ull;
    const unsigned long long nb = S.bytes[sg];
    if (off >= nb) return;
    char* d = (char*)S.dst[sg] + off;
    const int kind = S.kind[sg];
    const bool whole = off + 16 <= nb && (((uintptr_t)S.dst[sg]) & 15u) == 0;
    if (kind == 2) {
        if (whole) *reinterpret_cast<uint4*>(d) = make_uint4(0u, 0u, 0u, 0u);
        else for (unsigned long long i = off; i < nb && i < off + 16; ++i) ((char*)S.dst[sg])[i] = 0;
    } else if (kind == 1) {                // four fp32 of destination = four int64 of source
        const int64_t* s = (const int64_t*)S.src[sg] + off / 4;
        float* o = (float*)d;
        for (int i = 0; i < 4 && off + 4 * i < nb; ++i) o[i] = (float)s[i];
    } else {
        const char* s = (const char*)S.src[sg] + off;
        if (whole && (((uintptr_t)S.src[sg]) & 15u) == 0) *reinterpret_cast<uint4*>(d) = *reinterpret_cast<const uint4*>(s);
        else for (unsigned long long i = 0; i < 16 && off + i < nb; ++i) d[i] = s[i];
    }
}

extern "C" int re_step_stage_inputs(uint32_t* state, uint32_t seed, int64_t step, double lr, double beta1, double beta2, int32_t n,
                                    void* const* dst, const void* const* src, const int64_t* bytes, const int32_t* kind, re_stream_t stream) {
    re_clear_error();
    if ((state && step < 1) || n < 0 || n > ST_MAX || (n > 0 && (!dst || !src || !bytes || !kind))) return RE_EINVAL;
    StageSegs S{};
    S.n = n;
    unsigned total = 0;
    for (int i = 0; i < n; ++i) {
        if (!dst[i] || bytes[i] < 0 || kind[i] < 0 || kind[i] > 2 || (kind[i] != 2 && !src[i]) || (kind[i] == 1 && (bytes[i] & 3))) return RE_EINVAL;
        if (bytes[i] > (1ll << 40)) return RE_EUNSUPPORTED;
        S.dst[i] = dst[i]; S.src[i] = src[i]; S.bytes[i] = (unsigned long long)bytes[i]; S.kind[i] = kind[i];
        S.first[i] = total;
        total += (unsigned)re_cdiv(bytes[i], ST_CHUNK);
    }
    S.first[n] = total;
    if (total == 0) total = 1;
    const double st = step < 1 ? 1.0 : (double)step;
    hipLaunchKernelGGL(step_stage_inputs_k, dim3(total), dim3(256), 0, (hipStream_t)stream, state, seed, (float)(lr / (1.0 - pow(beta1, st))),
                       (float)(1.0 / sqrt(1.0 - pow(beta2, st))), S);
    return re_launch_status();
}

extern "C" int re_adam_step_dev(float* p, const float* g, float* m, float* v, int64_t n, const float* hyper, double beta1, double beta2,
                                double eps, double weight_decay, re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (!p || !g || !m || !v || !hyper || n < 0) return RE_EINVAL;
    if ((n & 3) || ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15u) != 0)
        return RE_EUNSUPPORTED;
    hipLaunchKernelGGL(adam_vec4_dev, dim3(re_grid(n >> 2, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)g, (float4*)m,
                       (float4*)v, n >> 2, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), hyper, (float)eps,
                       (float)weight_decay);
    return re_launch_status();
}

extern "C" int re_adam_step(float* p, const float* g, float* m, float* v, int64_t n, int64_t step, double lr, double beta1,
                            double beta2, double eps, double weight_decay, re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (!p || !g || !m || !v || n < 0 || step < 1) return RE_EINVAL;
    if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
          reinterpret_cast<uintptr_t>(v)) & 15u) != 0)
        return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const double bc1 = 1.0 - pow(beta1, (double)step);
    const double bc2 = 1.0 - pow(beta2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    const int64_t n4 = n >> 2;
    if (n4 > 0)
        hipLaunchKernelGGL(adam_vec4, dim3(re_grid(n4, 256)), dim3(256), 0, s, (float4*)p, (const float4*)g, (float4*)m, (float4*)v, n4,
                           (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), step_size, inv_sqrt_bc2, (float)eps, (float)weight_decay);
    if (n & 3)
        hipLaunchKernelGGL(adam_tail, dim3(1), dim3(256), 0, s, p, g, m, v, n4 << 2, n, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                           (float)(1.0 - beta2), step_size, inv_sqrt_bc2, (float)eps, (float)weight_decay);
    return re_launch_status();
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// clip_grad_norm_(parameters, max_norm) + Adam as two reductions and the optimizer launches (DeepFM/main.py:264-268: backward,
// nn.utils.clip_grad_norm_(.., 10), optimizer.step()): re_grad_clip_coef leaves coef = min(1, max_norm / (||g|| + 1e-6)) in a device word
// (torch's formula), re_adam_step_scaled / _dev_scaled multiply the gradient by it on the way in and write the clipped gradient back (what
// p.grad holds after the reference's step).  The norm: per-block partial sums of squares in a fixed order (thread-sequential over a strided
// slice, wave butterfly, four waves in order), then one wave over the partials -- deterministic.
#define SQN_BLOCKS 256
__global__ __launch_bounds__(256) void sqnorm_partial_k(const float4* __restrict__ g, int64_t n4, const float* __restrict__ tail, int ntail,
                                                        float* __restrict__ partial) {
    __shared__ float sw[4];
    float a = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = g[i];
        a = fmaf(v.x, v.x, a); a = fmaf(v.y, v.y, a); a = fmaf(v.z, v.z, a); a = fmaf(v.w, v.w, a);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) a = fmaf(tail[threadIdx.x], tail[threadIdx.x], a);
    a = re_wave_sum(a);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((sw[0] + sw[1]) + sw[2]) + sw[3];
}
__global__ __launch_bounds__(64) void clip_coef_k(const float* __restrict__ partial, int nb, float max_norm, float* __restrict__ out) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += 64) a += partial[i];
    a = re_wave_sum(a);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(a);
        const float c = max_norm / (norm + 1e-6f);
        out[0] = c < 1.0f ? c : 1.0f;
        out[1] = norm;
    }
}

extern "C" size_t re_grad_clip_workspace_bytes(void) { return SQN_BLOCKS * sizeof(float); }
// coef_norm[0] = min(1, max_norm / (||g||_2 + 1e-6)), coef_norm[1] = ||g||_2 over the flat gradient g[0 .. n)
extern "C" int re_grad_clip_coef(const float* g, int64_t n, float max_norm, float* coef_norm, void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!g || !coef_norm || n < 0 || !(max_norm > 0.f)) return RE_EINVAL;
    if (!ws || ws_bytes < re_grad_clip_workspace_bytes()) return RE_EWORKSPACE;
    if (reinterpret_cast<uintptr_t>(g) & 15u) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n >> 2;
    int nb = (int)re_cdiv(n4 > 0 ? n4 : 1, 1024);
    if (nb > SQN_BLOCKS) nb = SQN_BLOCKS;
    hipLaunchKernelGGL(sqnorm_partial_k, dim3(nb), dim3(256), 0, s, (const float4*)g, n4, g + (n4 << 2), (int)(n & 3), (float*)ws);
    hipLaunchKernelGGL(clip_coef_k, dim3(1), dim3(64), 0, s, (const float*)ws, nb, max_norm, coef_norm);
    return re_launch_status();
}

// Adam on gscale[0] * g; the scaled gradient is written back to g.  hyper == NULL: step_size / inv_sqrt_bc2 as given.
__global__ __launch_bounds__(256) void adam_vec4_scaled(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v,
                                                        int64_t n4, float b1, float b2, float omb1, float omb2, float step_size, float inv_sqrt_bc2,
                                                        const float* __restrict__ hyper, float eps, float wd, const float* __restrict__ gscale) {
    if (hyper) { step_size = hyper[0]; inv_sqrt_bc2 = hyper[1]; }
    if (inv_sqrt_bc2 == 0.f) return;   // (a gated step: see adam_vec4_dev)
    const float c = gscale[0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
        G.x *= c; G.y *= c; G.z *= c; G.w *= c;
        g[i] = G;
#define RE_ADAM1(c_) re_adam1(P.c_, M.c_, V.c_, G.c_, b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd);
        RE_ADAM1(x) RE_ADAM1(y) RE_ADAM1(z) RE_ADAM1(w)
#undef RE_ADAM1
        p[i] = P; m[i] = M; v[i] = V;
    }
}

// clip_grad_norm_(.., max_norm) + Adam over an arena of TWO weight-decay groups ([0, n_first) and the rest) as one launch behind the
// square-norm partials: every workgroup adds the partials itself (the same fixed order everywhere: the same coefficient), scales the gradient
// on its way through (written back: p.grad after the reference's step) and updates; workgroup 0 leaves [coefficient, norm] in coef_norm.
// (As re_grad_clip_coef + two re_adam_step_scaled this was four dispatches.)
__global__ __launch_bounds__(256) void adam_vec4_clip2(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v,
                                                       int64_t n4, int64_t n4_first, float b1, float b2, float omb1, float omb2, float step_size,
                                                       float inv_sqrt_bc2, const float* __restrict__ hyper, float eps, float wd_first, float wd_rest,
                                                       const float* __restrict__ partial, int nb, float max_norm, float* __restrict__ coef_norm) {
    __shared__ float sw[4];
    if (hyper) { step_size = hyper[0]; inv_sqrt_bc2 = hyper[1]; }
    float a = 0.f;
    for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
    a = re_wave_sum(a);
    if ((threadIdx.x & 63) == 0) sw[threadIdx.x >> 6] = a;
    __syncthreads();
    const float norm = sqrtf(((sw[0] + sw[1]) + sw[2]) + sw[3]);
    float c = max_norm / (norm + 1e-6f);
    c = c < 1.0f ? c : 1.0f;
    if (blockIdx.x == 0 && threadIdx.x == 0 && coef_norm) { coef_norm[0] = c; coef_norm[1] = norm; }
    if (inv_sqrt_bc2 == 0.f) return;   // (a gated step: see adam_vec4_dev)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i];
        G.x *= c; G.y *= c; G.z *= c; G.w *= c;
        g[i] = G;
        const float wd = i < n4_first ? wd_first : wd_rest;
#define RE_ADAM1(c_) re_adam1(P.c_, M.c_, V.c_, G.c_, b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd);
        RE_ADAM1(x) RE_ADAM1(y) RE_ADAM1(z) RE_ADAM1(w)
#undef RE_ADAM1
        p[i] = P; m[i] = M; v[i] = V;
    }
}

// n and n_first multiples of 4; step >= 1: the host's bias corrections, step == 0: `hyper` (device words).  ws: re_grad_clip_workspace_bytes().
extern "C" int re_adam_step_clip2(float* p, float* g, float* m, float* v, int64_t n, int64_t n_first, int64_t step, double lr, const float* hyper,
                                  double beta1, double beta2, double eps, double wd_first, double wd_rest, float max_norm, float* coef_norm,
                                  void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (!p || !g || !m || !v || n < 0 || n_first < 0 || n_first > n || (step < 1 && !hyper) || !(max_norm > 0.f)) return RE_EINVAL;
    if (!ws || ws_bytes < re_grad_clip_workspace_bytes()) return RE_EWORKSPACE;
    if ((n & 3) || (n_first & 3) || ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                                      reinterpret_cast<uintptr_t>(v)) & 15u) != 0)
        return RE_EUNSUPPORTED;
    float step_size = 0.f, inv_sqrt_bc2 = 1.f;
    if (step >= 1) {
        step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
        hyper = nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t n4 = n >> 2;
    int nb = (int)re_cdiv(n4, 1024);
    if (nb > SQN_BLOCKS) nb = SQN_BLOCKS;
    hipLaunchKernelGGL(sqnorm_partial_k, dim3(nb), dim3(256), 0, s, (const float4*)g, n4, (const float*)g, 0, (float*)ws);
    hipLaunchKernelGGL(adam_vec4_clip2, dim3(re_grid(n4, 256)), dim3(256), 0, s, (float4*)p, (float4*)g, (float4*)m, (float4*)v, n4, n_first >> 2,
                       (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), step_size, inv_sqrt_bc2, hyper, (float)eps, (float)wd_first,
                       (float)wd_rest, (const float*)ws, nb, max_norm, coef_norm);
    return re_launch_status();
}

// step >= 1: the host's bias corrections (re_adam_step's); step == 0: `hyper` (device words, re_adam_step_dev's).  n a multiple of 4.
extern "C" int re_adam_step_scaled(float* p, float* g, float* m, float* v, int64_t n, int64_t step, double lr, const float* hyper, double beta1,
                                   double beta2, double eps, double weight_decay, const float* gscale, re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (!p || !g || !m || !v || !gscale || n < 0 || (step < 1 && !hyper)) return RE_EINVAL;
    if ((n & 3) || ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15u) != 0)
        return RE_EUNSUPPORTED;
    float step_size = 0.f, inv_sqrt_bc2 = 1.f;
    if (step >= 1) {
        step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
        hyper = nullptr;
    }
    hipLaunchKernelGGL(adam_vec4_scaled, dim3(re_grid(n >> 2, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)p, (float4*)g, (float4*)m, (float4*)v,
                       n >> 2, (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), step_size, inv_sqrt_bc2, hyper, (float)eps,
                       (float)weight_decay, gscale);
    return re_launch_status();
}

// The owner's half of a data-parallel step (recboard_amd/dp.py): the gradient of this rank's slice of the arena arrives as `nparts` partial
// slices (one per rank, parts + r * part_stride: what an all-to-all of the ranks' gradient arenas leaves); g = gscale * (((part 0 + part 1) + ...)
// in rank order) -- every rank computes a slice exactly once, in a fixed order: the replicas stay bit-identical -- then Adam on the slice.
// g_out (optional): the reduced gradient.  nparts == 1, gscale == 1: re_adam_step's arithmetic on part 0.
__global__ __launch_bounds__(256) void adam_vec4_reduce(float4* __restrict__ p, const float4* __restrict__ parts, int nparts, int64_t stride4,
                                                        float4* __restrict__ g_out, float4* __restrict__ m, float4* __restrict__ v, int64_t n4, float b1,
                                                        float b2, float omb1, float omb2, float step_size, float inv_sqrt_bc2,
                                                        const float* __restrict__ hyper, float eps, float wd, float gscale) {
    if (hyper) { step_size = hyper[0]; inv_sqrt_bc2 = hyper[1]; }
    if (inv_sqrt_bc2 == 0.f) return;   // (a gated step: see adam_vec4_dev)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        float4 P = p[i], M = m[i], V = v[i], G = parts[i];
        for (int r = 1; r < nparts; ++r) {
            const float4 q = parts[(int64_t)r * stride4 + i];
            G.x += q.x; G.y += q.y; G.z += q.z; G.w += q.w;
        }
        if (gscale != 1.0f) { G.x *= gscale; G.y *= gscale; G.z *= gscale; G.w *= gscale; }
        if (g_out) g_out[i] = G;
#define RE_ADAM1(c_) re_adam1(P.c_, M.c_, V.c_, G.c_, b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd);
        RE_ADAM1(x) RE_ADAM1(y) RE_ADAM1(z) RE_ADAM1(w)
#undef RE_ADAM1
        p[i] = P; m[i] = M; v[i] = V;
    }
}
extern "C" int re_adam_step_reduce(float* p, const float* parts, int nparts, int64_t part_stride, float* g_out, float* m, float* v, int64_t n,
                                   int64_t step, double lr, const float* hyper, double beta1, double beta2, double eps, double weight_decay,
                                   double gscale, re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (!p || !parts || !m || !v || n < 0 || nparts < 1 || part_stride < n || (step < 1 && !hyper)) return RE_EINVAL;
    if ((n & 3) || (part_stride & 3) ||
        ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(parts) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v) |
          reinterpret_cast<uintptr_t>(g_out)) & 15u) != 0)
        return RE_EUNSUPPORTED;
    float step_size = 0.f, inv_sqrt_bc2 = 1.f;
    if (step >= 1) {
        step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
        inv_sqrt_bc2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
        hyper = nullptr;
    }
    hipLaunchKernelGGL(adam_vec4_reduce, dim3(re_grid(n >> 2, 256)), dim3(256), 0, (hipStream_t)stream, (float4*)p, (const float4*)parts, nparts,
                       part_stride >> 2, (float4*)g_out, (float4*)m, (float4*)v, n >> 2, (float)beta1, (float)beta2, (float)(1.0 - beta1),
                       (float)(1.0 - beta2), step_size, inv_sqrt_bc2, hyper, (float)eps, (float)weight_decay, (float)gscale);
    return re_launch_status();
}

// dst = alpha * src over a flat fp32 range (LightGCN: avgEmbds = allEmbds / (L+1), LightGCN/main.py:80)
__global__ __launch_bounds__(256) void scale_copy_k(float* __restrict__ dst, const float* __restrict__ src, float alpha, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) dst[i] = alpha * src[i];
}
extern "C" int re_scale_copy(float* dst, const float* src, float alpha, int64_t n, re_stream_t stream) {
    re_clear_error();
    if (n == 0) return RE_OK;
    if (!dst || !src || n < 0) return RE_EINVAL;
    hipLaunchKernelGGL(scale_copy_k, dim3(re_grid(n, 1024)), dim3(256), 0, (hipStream_t)stream, dst, src, alpha, n);
    return re_launch_status();
}
