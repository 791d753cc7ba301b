// Micro-benchmark: the score kernel's inner chain in isolation -- 32 dependent v_mfma_f32_32x32x2_f32 whose A operands come from
// LDS (8 x ds_read_b128 per lane) and whose B operands sit in registers -- cycles per chain, one wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define RSF 68
template <int MODE>   // 0: reads interleaved by the compiler, 1: all 8 reads before the chain, 2: no LDS reads (register operands)
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
    __shared__ __align__(16) float tile[64 * RSF];
    const int tid = threadIdx.x, lane = tid & 63, c = lane & 31, h = lane >> 5;
    for (int i = tid; i < 64 * RSF; i += 256) tile[i] = i * 1e-4f;
    float bq[32];
    for (int s = 0; s < 32; ++s) bq[s] = 1.0f + s * 0.01f + lane * 1e-3f;
    __syncthreads();
    f32x16 tot;
    for (int r = 0; r < 16; ++r) tot[r] = 0.f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        const float* arow = tile + ((i & 1) * 32 + c) * RSF + h * 32;
        f32x16 acc;
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        if (MODE == 1) {
            float4 a[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) a[q] = *reinterpret_cast<const float4*>(arow + 4 * q);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].x, bq[4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].y, bq[4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].z, bq[4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q].w, bq[4 * q + 3], acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                float4 a = MODE == 2 ? make_float4(bq[q], bq[q + 1], bq[q + 2], bq[q + 3]) : *reinterpret_cast<const float4*>(arow + 4 * q);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[4 * q + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[4 * q + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[4 * q + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[4 * q + 3], acc, 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) tot[r] += acc[r];
    }
    long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += tot[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE>
void run(const char* name, int blocks) {
    float* out; long long* cyc;
    (void)hipMalloc(&out, sizeof(float) * 256 * blocks); (void)hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s WGs=%d: %.0f cycles per 32-MFMA chain (2048 = back to back)\n", name, blocks, (double)c / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    for (int blocks : {256, 512}) {
        run<2>("register operands", blocks);
        run<0>("A from LDS, compiler's schedule", blocks);
        run<1>("A from LDS, all 8 reads before the chain", blocks);
    }
    return 0;
}
