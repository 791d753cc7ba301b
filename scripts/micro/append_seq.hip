// Micro-benchmark: cost of the score kernel's filter+append sequence (per score: v_cmp, ds_write2st64_b32, v_add, v_addc,
// v_lshl_add) on gfx950 -- with the LDS write, without it, and with the write but no dependency of its address on the chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#define A1(W) "v_cmp_ge_f32 vcc, %3, %4\n" W "v_add_u32 %2, 1, %2\n" "v_addc_co_u32 %0, vcc, 0, %0, vcc\n" "v_lshl_add_u32 %1, %0, 8, %5\n"
#define A4(W) A1(W) A1(W) A1(W) A1(W)
#define A16(W) A4(W) A4(W) A4(W) A4(W)
#define WR2 "ds_write2st64_b32 %1, %3, %2 offset0:0 offset1:112\n"
#define WRF "ds_write2st64_b32 %5, %3, %2 offset0:0 offset1:112\n"
#define WR1 "ds_write_b32 %1, %3\n"
#define WR64 "ds_write_b64 %1, %[pair]\n"
#define KERNEL(NAME, W)                                                                                   \
    __global__ void NAME(float* out, int iters, float thr) {                                              \
        extern __shared__ float lds[];                                                                    \
        int qn = 0;                                                                                       \
        const unsigned qbase = (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 28 * 256;                    \
        unsigned qaddr = qbase;                                                                           \
        int id = threadIdx.x;                                                                             \
        float x = threadIdx.x * 0.001f;                                                                   \
        double pr = x;                                                                                    \
        for (int i = 0; i < iters; ++i) {                                                                 \
            asm volatile(A16(W) : "+v"(qn), "+v"(qaddr), "+v"(id) : "v"(x), "v"(thr), "v"(qbase), [pair] "v"(pr) : "vcc", "memory"); \
            qn = 0; qaddr = qbase;                                                                        \
        }                                                                                                 \
        out[blockIdx.x * blockDim.x + threadIdx.x] = lds[threadIdx.x] + qn + id;                          \
    }
KERNEL(k_none, "")
KERNEL(k_w2, WR2)
KERNEL(k_wfix, WRF)
KERNEL(k_w1, WR1)
KERNEL(k_w64, WR64)

template <typename K>
void run(const char* name, K kern, int wgs_per_cu) {
    float* out;
    const int blocks = 256 * wgs_per_cu, threads = 256, iters = 20000;
    hipMalloc(&out, sizeof(float) * threads * blocks);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 64 * 1024, 0, out, 50, 1e30f);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 64 * 1024, 0, out, iters, 1e30f);
    hipEventRecord(b); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ns = ms * 1e6 / iters;
    printf("%-40s %d wave(s)/SIMD: %.1f ns per 16 scores per wave slot = %.0f cycles @2.4 GHz per score-step per SIMD\n", name, wgs_per_cu, ns,
           ns * 2.4 / 16 / wgs_per_cu);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 2; ++w) {
        run("no LDS write (4 VALU)", k_none, w);
        run("ds_write2st64_b32, chained address", k_w2, w);
        run("ds_write2st64_b32, fixed address", k_wfix, w);
        run("ds_write_b32, chained address", k_w1, w);
        run("ds_write_b64, chained address", k_w64, w);
    }
    return 0;
}
