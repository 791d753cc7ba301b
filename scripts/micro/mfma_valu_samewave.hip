// Micro-benchmark: how many VALU instructions of the SAME wave fit, for free, between the dependent 32x32x2 fp32 MFMAs of a
// chain (gfx950)?  A 32-MFMA chain is 2048 matrix-pipe cycles; G VALU ops (independent of the chain) are placed after every
// MFMA in program order by one asm block, so the compiler cannot re-cluster them.  dep=1: the G ops form one dependent chain;
// dep=0: they rotate over four independent registers.  Prints cycles per chain at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define V1(d, s) "v_max_f32 " d ", " s ", " d "\n"
#define REP0
#define REP2 V1("%1", "%5") V1("%2", "%5")
#define REP4 REP2 V1("%3", "%5") V1("%4", "%5")
#define REP6 REP4 REP2
#define REP8 REP4 REP4
#define REP10 REP8 REP2
#define REP12 REP8 REP4
#define REP16 REP8 REP8
#define DEP2 V1("%1", "%5") V1("%1", "%5")
#define DEP4 DEP2 DEP2
#define DEP6 DEP4 DEP2
#define DEP8 DEP4 DEP4
#define DEP10 DEP8 DEP2
#define DEP12 DEP8 DEP4
#define DEP16 DEP8 DEP8
#define MF "v_mfma_f32_32x32x2_f32 %0, %6, %7, %0\n"
#define STEP(R) MF R
#define CH4(R) STEP(R) STEP(R) STEP(R) STEP(R)
#define CH32(R) CH4(R) CH4(R) CH4(R) CH4(R) CH4(R) CH4(R) CH4(R) CH4(R)

#define KERNEL(NAME, R)                                                                                                  \
    __global__ void NAME(float* out, int iters) {                                                                        \
        f32x16 acc;                                                                                                      \
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;                                                                       \
        float a = threadIdx.x, b = a + 1, c = a + 2, d = a + 3, s = 0.5f * threadIdx.x;                                  \
        float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;                                                 \
        for (int i = 0; i < iters; ++i) {                                                                                \
            asm volatile(CH32(R) : "+v"(acc), "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(s), "v"(x), "v"(y));              \
        }                                                                                                                \
        float t = a + b + c + d;                                                                                         \
        for (int r = 0; r < 16; ++r) t += acc[r];                                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = t;                                                                  \
    }
KERNEL(k0, REP0)
KERNEL(k4, REP4) KERNEL(k6, REP6) KERNEL(k8, REP8) KERNEL(k10, REP10) KERNEL(k12, REP12) KERNEL(k16, REP16)
KERNEL(d4, DEP4) KERNEL(d6, DEP6) KERNEL(d8, DEP8) KERNEL(d10, DEP10) KERNEL(d12, DEP12) KERNEL(d16, DEP16)

template <typename K>
void run(const char* name, K kern, int g, int wgs_per_cu) {
    float* out;
    const int blocks = 256 * wgs_per_cu, threads = 256, iters = 2000;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, 50);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double us = ms * 1e3 / iters;
    printf("%-4s G=%2d VALU/MFMA (%4d per chain), %d wave(s)/SIMD: %.3f us per chain per wave slot; MFMA pipe busy %.0f %% (2048 cyc @2.4 GHz = 0.853 us)\n",
           name, g, 32 * g, wgs_per_cu, us, 100.0 * wgs_per_cu * 0.8533 / us);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) {
        run("ind", k0, 0, w); run("ind", k4, 4, w); run("ind", k6, 6, w); run("ind", k8, 8, w); run("ind", k10, 10, w);
        run("ind", k12, 12, w); run("ind", k16, 16, w);
        run("dep", d4, 4, w); run("dep", d6, 6, w); run("dep", d8, 8, w); run("dep", d10, 10, w); run("dep", d12, 12, w);
        run("dep", d16, 16, w);
    }
    return 0;
}
