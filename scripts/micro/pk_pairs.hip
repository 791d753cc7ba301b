// Stand-alone attempt at the round-6 finding (profiles/r6_handover_notes.txt): packed-fp32 arithmetic on register pairs fed by an LDS pair exchange, as
// the tile kernel's LayerNorm backward has it -- 4-wave workgroups, TWO per CU (58 KB of LDS) against ONE (84 KB) -- compiled twice: with v_pk_*_f32 and
// (target feature off) without.  Every launch's result is compared with the first launch's and with the other build's.
//   hipcc --offload-arch=gfx950 -O3 scripts/micro/pk_pairs.hip -o /tmp/pk_pairs && /tmp/pk_pairs
// OUTCOME (round 6, one MI355X, gpurun_out/r6_pk_pairs.txt): NOT reproduced -- 0 differing words in 20 launches of either build at one and at two workgroups
// per CU.  The instruction pattern alone is not enough; the reproducer that exists is the tile kernel itself: `make -C recboard_amd/csrc twopk` and
// `RE_TILE_LDS_KB=58 python scripts/hbm_poison_check.py --lib twopk --mode flush` (5 - 6 distinct results in 6 replays; RE_TILE_LDS_KB=84: one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define BODY                                                                                                               \
    extern __shared__ float lds[];                                                                                         \
    const int lane = threadIdx.x & 63, s = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;                                 \
    f2 d01 = in[(blockIdx.x * 256 + threadIdx.x) * 4 + 0], d23 = in[(blockIdx.x * 256 + threadIdx.x) * 4 + 1];             \
    f2 x01 = in[(blockIdx.x * 256 + threadIdx.x) * 4 + 2], x23 = in[(blockIdx.x * 256 + threadIdx.x) * 4 + 3];             \
    f2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};                                                                              \
    for (int it = 0; it < iters; ++it) {                                                                                   \
        float s1 = (d01.x + d01.y) + (d23.x + d23.y), s2 = fmaf(d01.x, x01.x, fmaf(d01.y, x01.y, fmaf(d23.x, x23.x, d23.y * x23.y))); \
        s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64); \
        float* b = lds + (it & 1) * 512;                                                                                   \
        if (g == 0) *reinterpret_cast<f2*>(b + (s * 16 + c) * 2) = (f2){s1, s2};                                           \
        __syncthreads();                                                                                                   \
        const f2 p0 = *reinterpret_cast<f2*>(b + (0 * 16 + c) * 2), p1 = *reinterpret_cast<f2*>(b + (1 * 16 + c) * 2);      \
        const f2 p2 = *reinterpret_cast<f2*>(b + (2 * 16 + c) * 2), p3 = *reinterpret_cast<f2*>(b + (3 * 16 + c) * 2);      \
        const f2 t = ((p0 + p1) + (p2 + p3)) * (1.0f / 64);                                                                \
        const f2 dx01 = rstd * (d01 - t.x - x01 * t.y), dx23 = rstd * (d23 - t.x - x23 * t.y);                             \
        acc01 += dx01; acc23 += dx23;                                                                                      \
        d01 = d01 * 0.999f + dx01 * 0.001f; d23 = d23 * 0.999f + dx23 * 0.001f;                                            \
    }                                                                                                                      \
    out[(blockIdx.x * 256 + threadIdx.x) * 2 + 0] = acc01; out[(blockIdx.x * 256 + threadIdx.x) * 2 + 1] = acc23;
__global__ __launch_bounds__(256, 2) void k_pk(const f2* in, f2* out, int iters, float rstd) { BODY }
__global__ __launch_bounds__(256, 2) __attribute__((target("no-packed-fp32-ops"))) void k_plain(const f2* in, f2* out, int iters, float rstd) { BODY }
int main() {
    const int G = 512, N = G * 256;
    std::vector<float> h(N * 8);
    unsigned r = 12345u;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = ((r >> 8) & 0xFFFF) / 65536.0f - 0.5f; }
    f2 *in, *out;
    (void)hipMalloc(&in, N * 8 * 4); (void)hipMalloc(&out, N * 4 * 4);
    (void)hipMemcpy(in, h.data(), N * 8 * 4, hipMemcpyHostToDevice);
    std::vector<float> refs[2] = {std::vector<float>(N * 4), std::vector<float>(N * 4)}, got(N * 4);
    for (int kb : {84, 58})
        for (int pk = 0; pk < 2; ++pk) {
            auto kern = pk ? k_pk : k_plain;
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
            long bad = 0, bad48 = 0;
            for (int rep = 0; rep < 20; ++rep) {
                hipLaunchKernelGGL(kern, dim3(G), dim3(256), kb * 1024, 0, in, out, 2000, 1.25f);
                (void)hipMemcpy(got.data(), out, N * 4 * 4, hipMemcpyDeviceToHost);
                if (kb == 84 && rep == 0) refs[pk] = got;
                const std::vector<float>& ref = refs[pk];
                for (int i = 0; i < N * 4; ++i)
                    if (memcmp(&got[i], &ref[i], 4)) { ++bad; bad48 += ((i / 4) & 63) >= 48; }
            }
            printf("LDS %d KB (%s per CU), %s: %ld words differ from the same build's first one-per-CU launch over 20 launches (%ld of them in lanes 48 - 63)\n", kb,
                   kb == 84 ? "one" : "two", pk ? "packed fp32" : "plain fp32 ", bad, bad48);
        }
    return 0;
}
