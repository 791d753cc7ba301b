// Micro-benchmark: achieved issue rate of fp32 MFMAs on gfx950 -- dependent chain vs independent accumulators, 1..4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float* out, long long* cyc, int iters) {
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int NACC>
__global__ void k16(float* out, long long* cyc, int iters) {
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 4; ++r) acc[a][r] = 0.f;
    float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc[a], 0, 0, 0);
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a)
        for (int r = 0; r < 4; ++r) s += acc[a][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <class K>
void run(const char* name, K kern, int nacc, int threads, int blocks, float flop_per_mfma) {
    float* out; long long* cyc;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 8 * nacc;              // MFMAs per wave
    const double waves = (double)threads / 64 * blocks;
    printf("%-28s acc=%d waves/WG=%d WGs=%d : %.1f cycles/MFMA/wave (timer), %.2f TFLOP/s total, %.3f ms\n", name, nacc, threads / 64, blocks,
           (double)c / nm, nm * waves * flop_per_mfma / (ms * 1e-3) / 1e12, ms);
    hipFree(out); hipFree(cyc);
}

int main() {
    // one WG per CU (256 CUs); waves per WG = 4 (1/SIMD), 8 (2/SIMD), 16 (4/SIMD)
    for (int threads : {256, 512, 1024}) {
        run("32x32x2 f32 dependent", k32<1>, 1, threads, 256, 4096.f);
        run("32x32x2 f32 2 chains", k32<2>, 2, threads, 256, 4096.f);
        run("16x16x4 f32 dependent", k16<1>, 1, threads, 256, 2048.f);
        run("16x16x4 f32 2 chains", k16<2>, 2, threads, 256, 2048.f);
        run("16x16x4 f32 4 chains", k16<4>, 4, threads, 256, 2048.f);
    }
    return 0;
}
