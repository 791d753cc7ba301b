// Micro-benchmark: do VALU instructions of one wave overlap with MFMAs of ANOTHER wave on the same SIMD (gfx950)?
// Each wave alternates a block of 32 dependent 32x32x2 MFMAs (2048 matrix-pipe cycles) with a block of NV dependent-ish
// VALU select ops.  1 wave/SIMD: the two blocks are serial by construction.  2 waves/SIMD: if the pipes overlap across
// waves, time per iteration approaches max(2*MFMA, 2*VALU) instead of 2*(MFMA+VALU).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV>
__global__ void k(float* out, int iters, int stagger) {
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    float l[16];
    for (int j = 0; j < 16; ++j) l[j] = (float)(16 - j) + threadIdx.x;
    float v = 3.5f + threadIdx.x;
    // optional phase offset for odd workgroups: start with the VALU block
    const bool odd = stagger && (blockIdx.x & 1);
    for (int i = 0; i < iters; ++i) {
        if (!odd || i > 0) {
#pragma unroll
            for (int u = 0; u < 32; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc, 0, 0, 0);
        }
#pragma unroll
        for (int rep = 0; rep < NV / 32; ++rep) {
#pragma unroll
            for (int j = 15; j >= 1; --j) {   // 2 VALU per slot: compare + select (sorted insertion of v into l)
                l[j] = (v > l[j - 1]) ? l[j - 1] : fmaxf(v, l[j]);
            }
            l[0] = fmaxf(v, l[0]);
            v += acc[0] * 1e-30f + 0.37f;
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[r] + l[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NV>
void run(int wgs_per_cu, int stagger) {
    float* out;
    const int blocks = 256 * wgs_per_cu, threads = 256;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    const int iters = 400;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(threads), 0, 0, out, 10, stagger);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<NV>, dim3(blocks), dim3(threads), 0, 0, out, iters, stagger);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("VALU block ~%4d ops, %d wave(s)/SIMD, stagger %d: %.1f us per iteration per wave-slot  (MFMA block alone = %.2f us at 2.1 GHz)\n", NV,
           wgs_per_cu, stagger, ms * 1e3 / iters, 2048 / 2100.0);
    hipFree(out);
}

int main() {
    run<32>(1, 0); run<32>(2, 0);
    run<512>(1, 0); run<512>(2, 0); run<512>(2, 1);
    run<1024>(1, 0); run<1024>(2, 0); run<1024>(2, 1);
    return 0;
}
