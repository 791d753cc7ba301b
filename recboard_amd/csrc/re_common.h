// Shared helpers for the recengine HIP kernels (gfx950 / CDNA4 only: wave64, no other target).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/recengine.h"

#define RE_WAVE 64

// hipGetLastError() is sticky per host thread and other libraries in the process (torch) may leave an error
// behind: every ABI entry clears it first, so re_launch_status() reports only this call's launches.
static inline void re_clear_error() { (void)hipGetLastError(); }
static inline int re_launch_status() {
    return hipGetLastError() == hipSuccess ? RE_OK : RE_ELAUNCH;
}

static inline int64_t re_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
#ifdef __HIPCC__
// Workgroup barrier that ALSO orders global memory between the workgroup's waves: every wave's vector-memory operations have completed
// before any wave passes.  `__syncthreads()` alone does not do that on gfx950: hipcc lowers its workgroup-scope release to `s_waitcnt
// lgkmcnt(0); s_barrier` (not tgsplit; the target's barrier does not drain the counters either), so a load behind the barrier can
// overtake another wave's store in front of it, and a flag stored behind it can overtake the data (round 5: the tile step's
// run-to-run differences with two workgroups per CU were exactly that).
__device__ __forceinline__ void re_sync_full() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
#endif
static inline size_t re_align(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// grid size for HBM-bound grid-stride kernels: enough blocks to fill 256 CUs x 8, never more than the work
static inline unsigned re_grid(int64_t work_items, int64_t per_block, int64_t cap = 2048) {
    int64_t g = re_cdiv(work_items, per_block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// Zero fill as a KERNEL.  hipMemsetAsync turns into a memset node when the stream is being captured, and a graph with such a
// node has been seen to go wrong from its second replay on (ROCm 7.2, gfx950: garbage in what the following kernels
// accumulate into the zeroed buffer); every entry point that may run inside a captured training step zeroes this way.
// `bytes` must be a multiple of 4 and `p` 4-byte aligned (all callers zero float / 32-bit word arrays).
static __global__ __launch_bounds__(256) void re_zero_k(uint32_t* __restrict__ p, size_t nwords) {
    const size_t stride = (size_t)gridDim.x * 256;
    if ((reinterpret_cast<uintptr_t>(p) & 15u) == 0) {
        uint4* p4 = reinterpret_cast<uint4*>(p);
        const size_t n4 = nwords >> 2;
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (size_t i = (n4 << 2) + (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = 0u;
    } else {
        for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride) p[i] = 0u;
    }
}
static inline hipError_t re_zero_async(void* p, size_t bytes, hipStream_t s) {
    if (bytes == 0) return hipSuccess;
    const size_t nwords = bytes >> 2;
    hipLaunchKernelGGL(re_zero_k, dim3(re_grid((int64_t)((nwords + 3) >> 2), 256)), dim3(256), 0, s, (uint32_t*)p, nwords);
    return hipGetLastError();
}

// Kernel arguments are read from the kernarg segment with scalar loads, and under scalar-register pressure the compiler RE-loads them where
// they are used instead of keeping them (enc_tail_k: 36 such loads all over the kernel): every first touch of a 64-byte line of the segment
// is then a memory round trip in the middle of a dependent chain (~5 k cycles each in the step's tail launch, scripts/tail_phases.py).
// re_kernarg_warm touches every line of the explicit arguments once, at the kernel's start, in one batch of scalar loads: the later loads
// hit the scalar cache.   usage: re_kernarg_warm<re_kernarg_bytes(&kernel<...>)>();
template <class... A>
__host__ __device__ constexpr int re_kernarg_bytes(void (*)(A...)) {
    int o = 0;
    ((o = (o + (int)alignof(A) - 1) / (int)alignof(A) * (int)alignof(A) + (int)sizeof(A)), ...);
    return o;
}
template <int BYTES, int FROM = 0>
__device__ __forceinline__ void re_kernarg_warm() {
    static_assert(BYTES > 0 && FROM % 64 == 0 && FROM < BYTES, "lines of the explicit arguments");
    if constexpr (BYTES - FROM > 12 * 64) re_kernarg_warm<BYTES, FROM + 12 * 64>();   // (twelve lines per batch)
    constexpr int LAST = (BYTES - 1) / 64 * 64;
#define RE_KA_OFF(k) (FROM + (k) * 64 < LAST ? FROM + (k) * 64 : LAST)
    typedef const uint32_t __attribute__((address_space(4))) ka_word;
    ka_word* ka = (ka_word*)__builtin_amdgcn_kernarg_segment_ptr();
    uint32_t a0, a1, a2, a3, a4, a5, a6, a7, a8, a9, a10, a11;
    // (one block: the loads' results arrive asynchronously, the wait has to sit in the same statement)
    asm volatile(
        "s_load_dword %0, %12, %13\n\ts_load_dword %1, %12, %14\n\ts_load_dword %2, %12, %15\n\ts_load_dword %3, %12, %16\n\t"
        "s_load_dword %4, %12, %17\n\ts_load_dword %5, %12, %18\n\ts_load_dword %6, %12, %19\n\ts_load_dword %7, %12, %20\n\t"
        "s_load_dword %8, %12, %21\n\ts_load_dword %9, %12, %22\n\ts_load_dword %10, %12, %23\n\ts_load_dword %11, %12, %24\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&s"(a0), "=&s"(a1), "=&s"(a2), "=&s"(a3), "=&s"(a4), "=&s"(a5), "=&s"(a6), "=&s"(a7), "=&s"(a8), "=&s"(a9), "=&s"(a10), "=&s"(a11)
        : "s"(ka), "i"(RE_KA_OFF(0)), "i"(RE_KA_OFF(1)), "i"(RE_KA_OFF(2)), "i"(RE_KA_OFF(3)), "i"(RE_KA_OFF(4)), "i"(RE_KA_OFF(5)), "i"(RE_KA_OFF(6)),
          "i"(RE_KA_OFF(7)), "i"(RE_KA_OFF(8)), "i"(RE_KA_OFF(9)), "i"(RE_KA_OFF(10)), "i"(RE_KA_OFF(11))
        : "memory");
#undef RE_KA_OFF
}

// One element of the dense Adam step with coupled weight decay.  The operation sequence is pinned (explicit fused / rounded operations):
// the same update is computed by adam_vec4 / adam_vec4_dev, by the scatter-add's row owners and by the gradient reduction's epilogue,
// and an eager step and a captured step must agree to the bit whichever of them runs.
__device__ __forceinline__ void re_adam1(float& p, float& m, float& v, float g, float b1, float b2, float omb1, float omb2, float step_size,
                                         float inv_sqrt_bc2, float eps, float wd) {
    const float gg = __fmaf_rn(wd, p, g);
    m = __fmaf_rn(b1, m, __fmul_rn(omb1, gg));
    v = __fmaf_rn(b2, v, __fmul_rn(__fmul_rn(omb2, gg), gg));
    p = __fsub_rn(p, __fmul_rn(step_size, __fdiv_rn(m, __fmaf_rn(__fsqrt_rn(v), inv_sqrt_bc2, eps))));
}

__device__ __forceinline__ float re_softplus(float x) {
    // log(1 + exp(x)), stable: max(x,0) + log1p(exp(-|x|))  (torch.nn.functional.softplus, threshold irrelevant in fp32 here)
    return fmaxf(x, 0.0f) + log1pf(expf(-fabsf(x)));
}
__device__ __forceinline__ float re_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float re_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
