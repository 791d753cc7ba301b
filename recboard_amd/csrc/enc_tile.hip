// K6/K7 training step of the SASRec encoder at D = 64: forward + criterion + backward of a 16-row tile by FOUR WAVES, one per
// 16-feature strip, with the tile's activations in registers (4 per lane and matrix) and every linear map on the XDL pipe.
//
// Reference restated: SASRec/main.py:178-193 (encode), :163-176 (after_one_block), :31-50 (PointWiseFeedForward), :199-215 (fit).
//
// Why this shape.  A Beauty-shaped batch of 512 is ~300 tiles: less than one wave of work per SIMD, so a launch lasts as long as ONE
// tile's dependent chain.  The workgroup-per-item kernel (enc_step.hip) spent 118 k cycles per item in ~75 barrier-separated phases
// over LDS tiles; one wave per tile with everything in registers (enc_wave.hip, the step before this one) has no barriers but is
// ONE in-order stream of ~20 k instructions at 4 - 5 cycles each.  Here a tile's stream is cut four ways by FEATURE STRIP: wave s
// owns features [16 s, 16 s + 16) of every activation, so all element-wise work (LayerNorm, dropout hashes, residuals, masks,
// operand splits, column sums) is a quarter per wave, each wave issues 6 MFMAs per linear map instead of 24 and holds 16 weight
// registers instead of 64, and what crosses waves is small: the bf16 planes of a map's input (4 KB per tile through LDS), per-token
// partial sums (LayerNorm, the criterion's dots) and the partial score tiles of the attention products.
//
// LAYOUTS (lane = 16 g + c, wave = 4 tile + strip s).  Of a tile matrix M[token][feature] a wave holds, in 4 registers,
//   T strip: m[j] = M[c][16 s + 4 g + j]      (the lane's token is c)         F strip: m[j] = M[4 g + j][16 s + c]   (the lane's feature)
// and a 16 x 16 score-type matrix S[query][key] as Rt: s[j] = S[c][4 g + j] or R: s[j] = S[4 g + j][c].
// v_mfma_f32_16x16x32_bf16 puts the OUTER index of both operands on lane & 15 and 8 k values on lane >> 4; with k slot i of step q
// standing for feature 16 (2 q + (i >> 2)) + 4 g + (i & 3), the four T strips of a matrix, lane by lane, ARE its two k-step
// fragments: the B operand of Y^T = W X^T (result: T strip s of Y from the weight rows of strip s as A) and the A operand of
// V = X Wv^T (result: F strip of V, which O^T = V^T P^T wants as A).  Weights are pre-arranged once per step in that k order, both
// orientations (enc_tile_prep_k).  Products over tokens or over the 16 features of a strip use v_mfma_f32_16x16x16_bf16: P V, dS K,
// dS^T Q, P^T dO strip-locally; q k^T and dO v^T as partial score tiles per strip, summed over the four waves in strip order.
// Arithmetic: operands are split hi = bf16(x), mid = bf16(x - hi); a product is hi.hi + hi.mid + mid.hi with fp32 accumulation
// (<= 3.01 * 2^-18 of sum |a b| per product: inside the 1e-4 parity bound); all row-wise arithmetic is fp32.
//
// WORK.  Workgroup = 4 waves = ONE tile of the plan (blockIdx = compact tile number; what a tile is -- shared by short sequences,
// or tile t of a sequence of nt tiles -- is read off the plan's row map), so the ~300 tiles of a batch spread over every CU.
// Tiles of short sequences are independent.  A long sequence's tile t attends to the keys of tiles 0..t, which other WORKGROUPS
// compute: per block, a tile publishes its k, v rows (forward) and its partial dK, dV for earlier tiles (backward) with device-scope
// stores, drains them (s_waitcnt vmcnt(0)), barriers, and one lane stores the launch's EPOCH into the tile's flag word; a consumer
// polls that word (device-scope load, bounded), barriers, and reads the rows with device-scope loads (MI355X guide: "handoff-flag").
// A forward wait is for a LOWER block index and a backward wait for tiles of the same sequence; the plan lays the long sequences'
// tiles first, so all of them are resident from the start as long as there are fewer of them than resident workgroups -- the plan
// kernel checks exactly that (hdr[7] = 1) and otherwise leaves the step to the workgroup-per-item kernel (enc_step.hip), which is
// launched right behind this one and returns at once when this one has run.  Sums over tiles are taken in tile order: deterministic.
// The weight gradients stay in enc_wgrad.hip (tape X, A, O, Y, HR + the six dY arrays); bias / LayerNorm gradients are column
// sums over a strip's tokens, written per TILE to the slab.
#include <math.h>

// NO PACKED-FP32 VECTOR INSTRUCTIONS IN THIS FILE'S DEVICE CODE (round 6; profiles/r6_handover_notes.txt): the Makefile compiles this file with
// `-Xclang -target-feature -Xclang -packed-fp32-ops`.  With `v_pk_mul_f32` / `v_pk_add_f32` / `v_pk_fma_f32` in the tile kernels (hipcc emits 336 of
// them at D = 64) and TWO workgroups resident per CU -- two waves per SIMD -- the LOW register of a packed result occasionally comes out wrong
// in its last sixteen lanes (the lane group of features 12 - 15, register 0 or 2 of the backward's incoming gradient: gradient-tape array dO2,
// columns 16 s + 12 / 16 s + 14; every later array of the tile follows).  Same code, same data: bit-identical with one workgroup per CU (84 KB
// of LDS requested), different from replay to replay inside one process with two (58 - 79 KB); not HBM contents, not memory waits
// (`-amdgpu-waitcnt-forcezero`: still random), not flat atomics, not VGPR-index mode, not the permlane swaps -- and gone (36 / 36 replays
// identical) once the compiler is told the target has no packed-fp32 operations.  (A `target("no-packed-fp32-ops")` attribute on the kernels
// does not compile: "illegal VGPR to SGPR copy"; the per-file flag does.)  Cost: +5 % vector instructions on a latency-bound chain.
#include "enc_fwd_item.h"
#include "enc_tile_prep.h"

#ifdef TL_PROFILE
#define TL_MARKS 96
#define TL_MARK_BLOCKS 8
__device__ unsigned long long g_tile_marks[TL_MARK_BLOCKS * TL_MARKS];   // the stamps of workgroups 0 .. 7 (the tiles of the first long sequences)
extern "C" int re_dbg_enc_marks_wave(unsigned long long* out) {          // (workgroup 0's)
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_marks), sizeof(unsigned long long) * TL_MARKS) == hipSuccess ? 0 : 1;
}
extern "C" int re_dbg_enc_marks_blocks(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tile_marks), sizeof(unsigned long long) * TL_MARK_BLOCKS * TL_MARKS) == hipSuccess ? 0 : 1;
}
#endif

#ifdef TL_HANDOVER_DEBUG
// Hand-over diagnostic build (`make hov` -> librecengine_hov.so; scripts/handover_soak.py): checksums of everything that crosses workgroups,
// the fenced / sc1-only protocol and the workgroups per CU selectable at run time.
#define TL_CHK_WORDS 24                                  // per tile: [2 l] k, [2 l + 1] v of block l; [8 + 3 l + slot] the inbox partials of block l
#define TL_CHK_TILES 65536
__device__ unsigned g_tl_chk[TL_CHK_TILES * TL_CHK_WORDS];
__device__ unsigned g_tl_stale[8];                       // [0..2] mismatches: forward k / v, backward k / v, dK / dV inbox; [4..6] checks made
__device__ int g_tl_fenced;
__device__ int g_tl_lds_total;                           // floats of LDS the launch requested (the part behind tl_lds_floats is a canary in this build)
__device__ unsigned g_tl_paranoid;                       // bit 0: a workgroup barrier in front of every operand-slot write; bit 1: ... of every exchange write
__device__ unsigned g_tl_fill;                           // != 0: every workgroup first fills its LDS with this bit pattern (does anything read LDS it has not written?)
static int h_tl_lds_kb = 84;
#define TL_FENCED (::g_tl_fenced != 0)
#define TL_PARANOID(BIT) do { if (::g_tl_paranoid & (BIT)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } while (0)
// bit 3: 8 wait states behind every 16-byte global store (is its data still being read when the next instruction overwrites the registers?)
#define TL_STORE_PAD() do { if (::g_tl_paranoid & 8u) asm volatile("s_nop 7" ::: "memory"); } while (0)
// bit 2: 16 extra wait states behind the last MFMA of every product, in front of anything that reads its result
#define TL_MFMA_PAD(ACC) do { if (::g_tl_paranoid & 4u) asm volatile("s_nop 15" : "+v"(ACC)); } while (0)
extern "C" int re_dbg_tile_handover(int fenced, int lds_kb) {
    h_tl_lds_kb = lds_kb;
    const int tot = lds_kb * 256;                        // (floats; D = 64, L <= 2 needs 57 984 bytes: every lds_kb >= 57 is what the launch requests)
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tl_lds_total), &tot, sizeof(int)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_fenced), &fenced, sizeof(int)) == hipSuccess ? 0 : 1;
}
extern "C" int re_dbg_tile_paranoid(unsigned bits) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_paranoid), &bits, sizeof(unsigned)) == hipSuccess ? 0 : 1;
}
__device__ unsigned g_tl_delay;                          // != 0: odd workgroups wait this many shader cycles at their start, every fourth one half of it in front of every
                                                         // further tile it takes: WHICH workgroup runs which tile after which changes -- do the results? (scripts/tile_order_check.py)
__device__ unsigned g_tl_skew;                           // != 0: behind EVERY barrier of the kernel the waves of a workgroup wait 0 .. 3 x this many cycles (which wave waits how long
                                                         // rotates from barrier to barrier): anything that leans on the waves running in step, not on a barrier, breaks
extern "C" int re_dbg_tile_skew(unsigned cycles) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_skew), &cycles, sizeof(unsigned)) == hipSuccess ? 0 : 1;
}
extern "C" int re_dbg_tile_delay(unsigned cycles) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_delay), &cycles, sizeof(unsigned)) == hipSuccess ? 0 : 1;
}
extern "C" int re_dbg_tile_fill(unsigned pattern) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_tl_fill), &pattern, sizeof(unsigned)) == hipSuccess ? 0 : 1;
}
extern "C" int re_dbg_tile_stale(unsigned* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tl_stale), 8 * sizeof(unsigned)) != hipSuccess) return 1;
    const unsigned z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    return (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_tl_stale), z, sizeof(z)) != hipSuccess) ? 1 : 0;
}
#else
#ifndef TL_FENCED
#define TL_FENCED 0
#endif
#define TL_PARANOID(BIT) do { } while (0)
#define TL_MFMA_PAD(ACC) do { } while (0)
#define TL_STORE_PAD() do { } while (0)
#endif

// The step kernels' ONE argument.  A field the tile uses once, late, is read from the kernel-argument segment where it is used (TL_ARG).
struct TlArgs {
    SeEmbed em;
    const int64_t* seq;
    int B, S, L;
    float drop_scale;
    uint32_t thresh, seed;
    float *u, *tape;
    EncTape T;
    const void* planp;
    EncHead H;
    float *dOut, *gtape, *slab;
    const uint32_t* seed_dev;
    float emb_scale;
    const uint32_t* wf;
    float* xch;
    int grid;                 // workgroups of the launch (the looped form hands out tiles beyond it)
};
template <class T>
__device__ __forceinline__ T tl_arg_at(unsigned off) {
    asm volatile("" : "+s"(off));
    typedef const char __attribute__((address_space(4))) ka_byte;
    typedef const T __attribute__((address_space(4))) ka_T;
    return *(ka_T*)((ka_byte*)__builtin_amdgcn_kernarg_segment_ptr() + off);
}
#define TL_ARG(M) tl_arg_at<decltype(((TlArgs*)nullptr)->M)>((unsigned)offsetof(TlArgs, M))

namespace tl4 {
#define TL_NS 4
#ifdef TL_L1INV
#define TL_L1INV_SYNCTHREADS 1
#endif
#include "enc_tile_body.inc"
#undef TL_NS
}   // namespace tl4
namespace tl8 {
#define TL_NS 8
#include "enc_tile_body.inc"
#undef TL_NS
}   // namespace tl8

template <int NS, typename KP, typename KS>
static int tl_launch(KP prep_k, KS step_k, const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P, float ds,
                     uint32_t thresh, uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan, int grid, const EncHead& H,
                     float* dx0, float* gtape, float* slab, float scale, uint32_t* wf, float* xch, int prep, hipStream_t s) {
    constexpr int D = 16 * NS;
    const EncTape T = enc_tape_layout(B, S, D, L);
    if (prep) {   // (0: the batch preparation launch of this step has written the fragments and advanced the epoch)
        hipLaunchKernelGGL(prep_k, dim3((unsigned)(TLC_PREP_THREADS(L, NS) / 256)), dim3(256), 0, s, P, (int)L, wf, enc_tile_epoch(tape, B, S, L, D));
        if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    }
    // ONE workgroup per CU (enc_tile_wg_per_cu): more than half of a CU's 160 KB of LDS is requested.  With two per CU (what 256 threads, ~190
    // registers and 58 KB allow at D = 64) the step's results differ from process to process -- one register of one wave, its last sixteen
    // lanes, usually in a tile a workgroup takes SECOND in the looped form -- although every launch is deterministic inside a process
    // (profiles/r5_handover_notes.txt: what round 5 ruled out; the kernel no longer spills scalar registers and drains vmcnt in front of the
    // barrier its waves exchange tape rows across).
    size_t ldsb = tl_lds_floats((int)L, NS) * sizeof(float);
#ifdef TL_HANDOVER_DEBUG
    if (ldsb < (size_t)h_tl_lds_kb * 1024) ldsb = (size_t)h_tl_lds_kb * 1024;

    if (grid > TL_CHK_TILES) return RE_EUNSUPPORTED;
#else
    if (enc_tile_wg_per_cu(16 * NS) == 1 && ldsb < (size_t)84 * 1024) ldsb = (size_t)84 * 1024;
#endif
#ifdef TL_LDS_KB_ENV   // (round-6 experiment builds only: the LDS request from the environment -- 84: ONE workgroup per CU whatever the grid)
    if (const char* e = getenv("RE_TILE_LDS_KB")) { const size_t want = (size_t)atoi(e) * 1024; if (want > ldsb) ldsb = want; }
#endif
    if (hipFuncSetAttribute((const void*)step_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
    const TlArgs A{em, seq, (int)B, (int)S, (int)L, ds, thresh, seed, u, (float*)tape, T, plan, H, dx0, gtape, slab, seed_dev, scale, (const uint32_t*)wf, xch, grid};
    hipLaunchKernelGGL(step_k, dim3(grid), dim3(64 * NS), ldsb, s, A);
    return hipGetLastError() == hipSuccess ? RE_OK : RE_ELAUNCH;
}

// D = 64: four waves per tile (namespace tl4); D = 128: eight (tl8).  The same body (enc_tile_body.inc).
int enc_tile_step_launch(int64_t D, const SeEmbed& em, const int64_t* seq, int64_t B, int64_t S, int64_t L, const SasrecParams& P, float ds, uint32_t thresh,
                         uint32_t seed, const uint32_t* seed_dev, float* u, void* tape, const void* plan, int grid, const EncHead& H, float* dx0,
                         float* gtape, float* slab, float scale, uint32_t* wf, float* xch, int prep, hipStream_t s) {
    const bool loop = enc_tile_looped(B, S);
    if (D == 64)
        return tl_launch<4>(tl4::enc_tile_prep_k, loop ? tl4::enc_tile_step_k<true> : tl4::enc_tile_step_k<false>, em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, grid, H, dx0, gtape,
                            slab, scale, wf, xch, prep, s);
    if (D == 128)
        return tl_launch<8>(tl8::enc_tile_prep_k, loop ? tl8::enc_tile_step_k<true> : tl8::enc_tile_step_k<false>, em, seq, B, S, L, P, ds, thresh, seed, seed_dev, u, tape, plan, grid, H, dx0, gtape,
                            slab, scale, wf, xch, prep, s);
    return RE_EUNSUPPORTED;
}
