// Fills (nearly) every vector register of every SIMD with one bit pattern and exits: does a later kernel read a register lane it never wrote?
// (scripts/vgpr_pollute_check.py: the SASRec tile step's results after pollute(0) and after pollute(NaN) must be the same bits.)
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/vgpr_pollute.hip -o gpurun_out/libpollute.so
#include <hip/hip_runtime.h>
#define NR 496
__global__ __launch_bounds__(64) void pollute_k(unsigned pattern, unsigned* sink) {
    unsigned r[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) { r[i] = pattern; asm volatile("" : "+v"(r[i])); }
    // (a second pass keeps all of them live at once)
    unsigned acc = 0;
#pragma unroll
    for (int i = 0; i < NR; ++i) { asm volatile("" : "+v"(r[i])); acc ^= r[i]; }
    if (acc == 0x12345u && sink) sink[0] = acc;   // (never true for the patterns used: the stores keep nothing alive but the chain)
}
extern "C" int pollute(unsigned pattern, void* stream) {
    hipLaunchKernelGGL(pollute_k, dim3(8192), dim3(64), 0, (hipStream_t)stream, pattern, (unsigned*)nullptr);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
