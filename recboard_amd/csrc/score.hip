// K4: full-catalog scoring  S = Q . E^T  on the fp32 matrix cores, with the Coach.evaluate epilogue fused
// (seen-mask + streaming top-K), so the B x N score matrix never exists.          MFMA-bound (fp32).
//
// Reference ops replaced: SASRec/main.py:228 einsum("BD,ND->BN"), MF-BPR/main.py:104, LightGCN/main.py:120,
// then -- in freerec's Coach.evaluate, mirrored at UniSRec/main.py:408-414 -- `scores[seen] = -1e23`, a dense
// target matrix and one torch.topk per "NAME@K" monitor.
//
// Arithmetic: v_mfma_f32_32x32x2_f32.  A = 32 items x 2 k, B = 2 k x 32 users, so a lane owns ONE user (its
// accumulator column) and 16 items (rows).  k is fed in natural order (step s uses k = 2s, 2s+1), which makes
// every score bit-for-bit the fp32 chain  acc = fmaf(q[k], e[k], acc), k = 0..D-1  -- the C oracle's arithmetic.
//
// Data movement per workgroup (256 threads = 4 waves = 4 x 32 users):
//   * the 128 users' query rows are MFMA-B fragments held in registers for a whole segment (D/2 VGPRs);
//   * item rows stream HBM/L2 -> registers (coalesced float4, prefetched one stage ahead) -> LDS, de-interleaved
//     into [even k | odd k] halves with a 16-B row pad so that the A-fragment ds_read_b128s are conflict-free;
//   * per 32-item tile a wave issues D/2 MFMAs (D=64: 2048 cycles) and then scans its 16 accumulators against
//     the lane's user threshold (the current K-th best): one v_cmp per register in the common case.
//   * rare hits go to a per-user binary heap in LDS ([slot][user] layout, conflict-free across lanes); the two
//     lanes that share a user insert one after the other.  The seen-mask is applied only to hits, through a
//     per-lane cursor into the user's sorted seen list.
// Work split: "stream-K" -- the (user block, item stage) units are cut into equal contiguous ranges, one per
// workgroup (2 workgroups per CU), so the chip is evenly loaded for any B x N; a user block touched by several
// workgroups gets one partial top-K list per segment, merged by score_topk_merge (bitonic networks on wave64).
//
// Algorithmic work: 2*D FLOP per (user, item) pair; bytes lower bound 4D(B+N) + 12BK.
#include <math.h>

#include "re_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SC_USERS 128   // users per workgroup (4 waves x 32)
#define SC_TI 64       // items per LDS stage (2 MFMA tiles)
#define SC_MAX_WGS 512 // 2 per CU on MI355X
#define SC_MIN_SEG 1   // stages (of SC_TI items) per segment, at least (A/B in scripts/tune_score.py: larger is slower)

__device__ __forceinline__ bool sc_before(float va, int ia, float vb, int ib) {  // a ranks before b
    return va > vb || (va == vb && ia < ib);
}

// ---------------------------------------------------------------------------------------------------------
template <int D, bool TOPK, int POPMODE>   // POPMODE 0: per register, 1: per-lane pop, 2: per tile by hit density
__global__ __launch_bounds__(256, 2) void score_kernel(const float* __restrict__ Q, const float* __restrict__ E,
                                                       int64_t B, int64_t N, const int64_t* __restrict__ seen_ptr,
                                                       const int64_t* __restrict__ seen_idx, int K,
                                                       float* __restrict__ part_vals, int* __restrict__ part_idx,
                                                       int maxseg, int64_t nub, int64_t nst, int64_t upw,
                                                       float* __restrict__ dense_out) {
    constexpr int KH = D / 2;            // k values per lane half
    constexpr int RSF = D + 4;           // LDS row stride in floats (16-B pad)
    constexpr int F4_PER_STAGE = SC_TI * D / 4;
    constexpr int PF = F4_PER_STAGE / 256;  // prefetch float4 per thread
    extern __shared__ __align__(16) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);
    float* hv = tile + SC_TI * RSF;
    int* hi = reinterpret_cast<int*>(hv + (TOPK ? K : 0) * SC_USERS);
    int* cnt = hi + (TOPK ? K : 0) * SC_USERS;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int ul = wid * 32 + c;  // user slot inside the block

    const int64_t units_total = nub * nst;
    int64_t unit = (int64_t)blockIdx.x * upw;
    const int64_t unit_end = (unit + upw < units_total) ? unit + upw : units_total;

    while (unit < unit_end) {
        const int64_t ub = unit / nst;
        const int64_t st0 = unit - ub * nst;
        const int64_t st1 = (st0 + (unit_end - unit) < nst) ? st0 + (unit_end - unit) : nst;
        const int seg = (int)((int64_t)blockIdx.x - (ub * nst) / upw);
        const int64_t user = ub * SC_USERS + ul;

        // ---- query fragments (MFMA B operand): bq[s] = Q[user][2s + h]
        float bq[KH];
        {
            const float* qrow = Q + user * D;
            const bool uok = user < B;
#pragma unroll
            for (int s = 0; s < KH; ++s) bq[s] = uok ? qrow[2 * s + h] : 0.0f;
        }
        float thr = -INFINITY;
        int64_t sc_cur = 0, sc_end = 0;
        int next_seen = 0x7FFFFFFF;
        if (TOPK) {
            if (h == 0) cnt[ul] = 0;
            if (seen_ptr && user < B) {
                sc_cur = seen_ptr[user];
                sc_end = seen_ptr[user + 1];
                // first seen id inside this segment
                const int64_t first_item = st0 * SC_TI;
                int64_t lo = sc_cur, hi2 = sc_end;
                while (lo < hi2) {
                    const int64_t mid = (lo + hi2) >> 1;
                    if (seen_idx[mid] < first_item) lo = mid + 1; else hi2 = mid;
                }
                sc_cur = lo;
                next_seen = sc_cur < sc_end ? (int)seen_idx[sc_cur] : 0x7FFFFFFF;
            }
        }

        // ---- prefetch first stage
        float4 pf[PF];
        auto prefetch = [&](int64_t st) {
            const int64_t item0 = st * SC_TI;
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int f = p * 256 + tid;
                const int row = f / (D / 4);
                pf[p] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (item0 + row < N) pf[p] = reinterpret_cast<const float4*>(E + item0 * D)[f];
            }
        };
        prefetch(st0);

        for (int64_t st = st0; st < st1; ++st) {
            __syncthreads();  // every wave is done reading the previous stage
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int f = p * 256 + tid;
                const int row = f / (D / 4);
                const int k0 = (f % (D / 4)) * 4;
                float* dst = tile + row * RSF + (k0 >> 1);
                *reinterpret_cast<float2*>(dst) = make_float2(pf[p].x, pf[p].z);        // even k
                *reinterpret_cast<float2*>(dst + KH) = make_float2(pf[p].y, pf[p].w);   // odd k
            }
            __syncthreads();
            if (st + 1 < st1) prefetch(st + 1);

#pragma unroll
            for (int it = 0; it < SC_TI / 32; ++it) {
                const int64_t item0 = st * SC_TI + it * 32;
                if (item0 >= N) break;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                const float* arow = tile + (it * 32 + c) * RSF + h * KH;
#pragma unroll
                for (int q = 0; q < KH / 4; ++q) {
                    const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
                    if (TOPK) {
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[4 * q + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[4 * q + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[4 * q + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[4 * q + 3], acc, 0, 0, 0);
                    } else {  // dense: users on rows, items on lanes -> coalesced row stores
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[4 * q + 0], a.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[4 * q + 1], a.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[4 * q + 2], a.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[4 * q + 3], a.w, acc, 0, 0, 0);
                    }
                }

                if (!TOPK) {
                    const int64_t item = item0 + c;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int64_t urow = ub * SC_USERS + wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (urow < B && item < N) dense_out[urow * N + item] = acc[r];
                    }
                    continue;
                }

                // ---- fast filter: one compare per accumulator register against the lane's user threshold
                unsigned m = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) m |= (acc[r] > thr ? 1u : 0u) << r;
                if (__ballot(m != 0) == 0ull) continue;

                // ---- slow path: heap inserts.
                // POP: every lane pops ITS OWN lowest pending hit per iteration (loop count = max over lanes of
                // popcount(m)); !POP: register after register (loop count = #registers with any hit in the wave).
                if (user >= B) m = 0;
                // few lanes with hits (warm lists): walking the registers is cheapest; many (cold lists, iid scores): pop
                const bool POP = POPMODE == 1 || (POPMODE == 2 && __popcll(__ballot(m != 0)) >= 12);
#pragma unroll 1
                for (int rr = 0; POP ? (__ballot(m != 0) != 0ull) : (rr < 16); ++rr) {
                    bool hit;
                    int r;
                    float v;
                    if (POP) {
                        hit = m != 0;
                        r = hit ? (__ffs(m) - 1) : 0;
                        m &= m - 1;
                        v = acc[0];
#pragma unroll
                        for (int q = 1; q < 16; ++q) v = (r == q) ? acc[q] : v;
                    } else {
                        r = rr;
                        hit = (m >> r) & 1u;
                        if (__ballot(hit) == 0ull) continue;
                        v = acc[r];
                    }
                    const int item = (int)item0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    // `>=`: thr may have been raised inside this tile by the partner lane with a HIGHER item id; the
                    // insert below applies the full (value, lowest-index) rule
                    hit = hit && item < N && v >= thr;
                    if (hit && item >= next_seen) {  // advance the seen cursor to lower_bound(item)
                        int64_t lo = sc_cur, hi2 = sc_end;
                        while (lo < hi2) {
                            const int64_t mid = (lo + hi2) >> 1;
                            if (seen_idx[mid] < item) lo = mid + 1; else hi2 = mid;
                        }
                        sc_cur = lo;
                        next_seen = sc_cur < sc_end ? (int)seen_idx[sc_cur] : 0x7FFFFFFF;
                        if (next_seen == item) hit = false;  // scores[seen] = -1e23 never reaches the top-K
                    }
#pragma unroll 1
                    for (int hh = 0; hh < 2; ++hh) {
                        if (hit && h == hh) {
                            const int n_in = cnt[ul];
                            if (n_in < K) {  // still filling: sift-up insert (root = worst entry)
                                int p = n_in;
                                while (p > 0) {
                                    const int par = (p - 1) >> 1;
                                    const float pv = hv[par * SC_USERS + ul];
                                    const int pi = hi[par * SC_USERS + ul];
                                    if (!sc_before(pv, pi, v, item)) break;  // parent not better than new -> stop
                                    hv[p * SC_USERS + ul] = pv;
                                    hi[p * SC_USERS + ul] = pi;
                                    p = par;
                                }
                                hv[p * SC_USERS + ul] = v;
                                hi[p * SC_USERS + ul] = item;
                                cnt[ul] = n_in + 1;
                            } else if (sc_before(v, item, hv[ul], hi[ul])) {  // beats the current worst: replace root
                                int p = 0;
                                for (;;) {
                                    int ch = 2 * p + 1;
                                    if (ch >= K) break;
                                    float cv = hv[ch * SC_USERS + ul];
                                    int ci = hi[ch * SC_USERS + ul];
                                    if (ch + 1 < K) {
                                        const float rv = hv[(ch + 1) * SC_USERS + ul];
                                        const int ri = hi[(ch + 1) * SC_USERS + ul];
                                        if (sc_before(cv, ci, rv, ri)) { cv = rv; ci = ri; ++ch; }  // pick the WORSE child
                                    }
                                    if (!sc_before(v, item, cv, ci)) break;  // new entry is not better than child -> it stays here
                                    hv[p * SC_USERS + ul] = cv;
                                    hi[p * SC_USERS + ul] = ci;
                                    p = ch;
                                }
                                hv[p * SC_USERS + ul] = v;
                                hi[p * SC_USERS + ul] = item;
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                    if (POP) thr = (cnt[ul] >= K) ? hv[ul] : -INFINITY;   // later pops of this tile see the tightened threshold
                }
                thr = (cnt[ul] >= K) ? hv[ul] : -INFINITY;
            }
        }

        if (TOPK) {
            // ---- partial list of this segment: unsorted heap contents, padded with (-inf, -1)
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (user < B) {
                const int n_in = cnt[ul];
                float* pv = part_vals + (user * maxseg + seg) * K;
                int* pi = part_idx + (user * maxseg + seg) * K;
                for (int s = h; s < K; s += 2) {
                    pv[s] = s < n_in ? hv[s * SC_USERS + ul] : -INFINITY;
                    pi[s] = s < n_in ? hi[s * SC_USERS + ul] : -1;
                }
            }
        }
        unit += st1 - st0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// Variant R ("register lists"): the same MFMA loop, but every LANE keeps its own best-K list of the items IT sees
// (lane (c,h) sees 16 of every 32 items of user c) -- sorted, in registers.  Hits are appended to a per-lane FIFO in LDS
// with plain stores (no atomics: the queue is private to the lane) and folded into the register list in wave-wide drains:
// round e processes entry e of every lane's queue with ONE branch-free sorted insertion, so the cost of a drain is
// max-over-lanes(queue length) insertions while all 64 lanes work -- instead of one ~440-cycle LDS heap episode per hit with
// one or two lanes active (the PMC finding of round 1).  A lane's K-th best is a valid lower bound of its user's K-th best,
// so filtering on it is exact; the two lanes of a user emit two partial lists per segment and score_topk_merge takes the
// best K of all of them.
//
// List entries are ONE sortable 64-bit key per (score, item): the score widened to double (exact) with the item id in the
// 29 mantissa bits a float does not have -- 2^29-1-id for scores >= 0, id for negative ones, so that "greater key" is
// exactly "greater score, or equal score and lower id" (the tie rule).  The sorted insertion of key k into slot j is then
// clamp(k, list[j], list[j-1]) = v_min_f64 + v_max_f64: no compare, no select, no mask hazards.  The drains are VALU
// bound (a (value, id) pair of lists costs 1 compare + 4 selects per slot at ~7 cycles per dependent op), and this is
// what the insertion cost is made of.  (Tried and dropped: value-only lists with the (value, item) pairs logged to global
// memory -- the end-of-segment selection over the log is latency bound.)
#define SR_QC 28
#define SR_VOTE 14  // a workgroup drain is called when some queue holds more than this at the top of a stage
// Filter + append of ONE score in 4 vector instructions: the score and its item id are written to the lane's queue tail
// unconditionally (one ds_write2st64), the tail advances by the compare bit (v_cmp -> v_addc carry-in); the running item id
// moves on to the next accumulator register in the two wait states a VALU-written VCC needs before a VALU reads it as carry.
// Operands: %0 qn, %1 qaddr, %2 id; %3 thr, %4 qbase, then the accumulator registers.  KM: optional "vcc &= ~kill mask".
#define SR_APPEND1(ACC, KM, DELTA)                                  \
    "v_cmp_ge_f32 vcc, " ACC ", %3\n"                               \
    KM                                                              \
    "ds_write2st64_b32 %1, " ACC ", %2 offset0:0 offset1:%c[qoff]\n" \
    "v_add_u32 %2, " #DELTA ", %2\n"                                \
    "v_addc_co_u32 %0, vcc, 0, %0, vcc\n"                           \
    "v_lshl_add_u32 %1, %0, 8, %4\n"
#define SR_TAGBITS 29
// raw v_min_f64 / v_max_f64: through fmin / fmax the compiler re-canonicalises every loop-carried list element
// (one extra v_max_f64 v, v, v per slot per insertion); keys are never NaN
__device__ __forceinline__ double sr_min(double a, double b) {
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// order-preserving float <-> uint32 (larger float <-> larger word; 0 is below every float): the shared per-user bound
__device__ __forceinline__ unsigned sr_enc(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sr_dec(unsigned e) {
    if (e == 0u) return -INFINITY;
    return __uint_as_float((e & 0x80000000u) ? (e ^ 0x80000000u) : ~e);
}
__device__ __forceinline__ double sr_max(double a, double b) {
    double r;
    asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ unsigned long long g_sr_counters[4];
__device__ unsigned long long g_sr_counters_x[2];   // diagnostics (dbg == 3): drains, rounds, appended hits, tiles with hits
// KT > 0: K is the compile-time constant KT (= KR): the K-th / (K/2)-th list slots are fixed registers instead of a 2 x KR
// select chain per drain (the common K = 50 of the evaluation monitors gets its own instantiation)
//
// X2 = true ("split" variant, score_topk's fast path): Q and E arrive as bf16 hi/mid planes (score_split_k: row = [D hi | D mid],
// 4D bytes like the fp32 row) and a score is three v_mfma_f32_32x32x16_bf16 products per 16 k -- hi.hi + hi.mid + mid.hi -- on
// the XDL matrix pipe, which (unlike the fp32 "SGEMM" MFMA) runs beside the vector ALU and costs 384 instead of 2048 cycles per
// 32 x 32 tile.  These scores are APPROXIMATE (|s' - s| <= eps_u, see score_topk_merge_x): the kernel selects candidates with
// them, the merge kernel re-scores the candidates with the exact fmaf chain and certifies the result (or flags the user for
// the exact kernel).  Everything else -- queues, lists, bounds, seen handling, stream-K split -- is shared with the exact form.
// blockflag != NULL (exact form as the fallback pass): user blocks whose flag is 0 are skipped.
// NB (split form): LDS stage buffers.  2: the next stage is requested right behind the stage barrier, the seen window and the
// bound word once per stage.  3 (long catalogs: the rows come from HBM and take longer to arrive than a stage takes to score): stages
// are requested TWO ahead, and nothing else is loaded per stage -- the memory counter retires in order, so a wait for any other load
// would drain the stage loads in flight: the seen window is reloaded only when a lane has used it up (blocking, rare: a few ids per
// user over thousands of stages), the bound word every 16th stage.  The third buffer costs 5 queue entries per lane at D = 64.
template <int D, int KR, int KT, bool X2, int NB = 2>
__global__ __launch_bounds__(256, D == 64 ? 2 : 1) void score_kernel_reg(const float* __restrict__ Q, const float* __restrict__ E,
                                                           int64_t B, int64_t N, const int64_t* __restrict__ seen_ptr,
                                                           const int64_t* __restrict__ seen_idx, int K,
                                                           float* __restrict__ part_vals, int* __restrict__ part_idx,
                                                           int maxseg, int64_t nub, int64_t nst, int64_t upw,
                                                           unsigned* __restrict__ gthr, int dbg,
                                                           const int* __restrict__ blockflag, float* __restrict__ part_T, int segs) {
    // Fallback pass with nobody flagged (the normal case): leave before anything else -- the kernel's prologue spills loop
    // invariants to scratch memory, 37 MB of writes per launch over 512 workgroups that an early exit further down does not avoid.
    if (blockflag && blockflag[nub + 1] == 0) return;
    constexpr int KH = D / 2;
    constexpr int RSF = D + 4;
    constexpr int NS16 = D / 16;   // X2: MFMA steps of 16 k
    constexpr int F4_PER_STAGE = SC_TI * D / 4;
    constexpr int PF = F4_PER_STAGE / 256;
    extern __shared__ __align__(16) unsigned char smem[];
    float* tile = reinterpret_cast<float*>(smem);
    // Split form: the item stages go global -> LDS directly (global_load_lds_dwordx4: no staging registers, whose spilling had
    // put a wait for the prefetch right behind its issue), double buffered, rows unpadded with their 16-byte chunks XOR-swizzled
    // by the row number (conflict-free fragment reads); the queues give up 8 entries per lane to make room for the second buffer.
    constexpr int QC = X2 ? (NB == 3 && D == 64 ? 15 : 20) : SR_QC;
    constexpr int TILE_FLOATS = X2 ? NB * SC_TI * D : SC_TI * RSF;
    static_assert(NB == 2 || (NB == 3 && X2), "three stage buffers: split form only");
    static_assert(!X2 || D == 64 || D == 128, "split form: D = 64 or 128");
    float* qv = tile + TILE_FLOATS;                          // [4 waves][QC][64 lanes]
    int* qi = reinterpret_cast<int*>(qv + 4 * QC * 64);
    __shared__ __align__(16) int vote[4];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 31, h = lane >> 5;
    const int ul = wid * 32 + c;
    const int NONE = 0x7FFFFFFF;
    const int vote_at = (dbg >> 8) ? (dbg >> 8) - 1 : (X2 ? (QC - 10 < 8 ? QC - 10 : 8) : SR_VOTE);   // (tuning override in the upper bits of dbg)
    dbg &= 0xFF;
    float* myqv = qv + wid * QC * 64 + lane;
    int* myqi = qi + wid * QC * 64 + lane;
    const unsigned long long TAGMASK = (1ull << SR_TAGBITS) - 1ull;
    // (score, item) -> sortable key; key -> score / item
    auto make_key = [&](float v, int it) -> double {
        v += 0.0f;   // -0 -> +0
        const unsigned long long tag = (v >= 0.0f) ? (TAGMASK - (unsigned long long)it) : (unsigned long long)it;
        return __longlong_as_double((long long)((unsigned long long)__double_as_longlong((double)v) | tag));
    };
    auto key_value = [&](double k) -> float {
        return (float)__longlong_as_double((long long)((unsigned long long)__double_as_longlong(k) & ~TAGMASK));
    };
    auto key_item = [&](double k) -> int {
        const unsigned long long b = (unsigned long long)__double_as_longlong(k);
        const unsigned long long tag = b & TAGMASK;
        return (int)((b >> 63) ? tag : TAGMASK - tag);
    };
    // empty slot: below every real entry (a real item id is < 2^29 - 1); decodes to (-FLT_MAX, no item)
    const double KEMPTY = __longlong_as_double((long long)((unsigned long long)__double_as_longlong((double)-3.402823466e+38f) | TAGMASK));

    // Work split.  segs == 0: stream-K (score_plan) -- the (user block, stage) units in one line, cut into equal ranges.
    // segs > 0 ("sliced", few user blocks: score_plan_topk): every user block's catalog is cut into the same `segs` slices of
    // upw stages and workgroup w takes slice 8 (j / nub) + (w mod 8) of user block j mod nub, j = w / 8 -- consecutive workgroup
    // ids go round the XCDs, so the workgroups of ONE XCD that run together score the SAME slice for different user blocks and
    // the slice comes out of that XCD's L2 for all but the first of them (512 users x 12.5 M items: the table was streamed from
    // HBM once per user block, 12.8 GB per call, and the kernel ran at HBM speed).
    const int64_t units_total = nub * nst;
    int64_t unit = (int64_t)blockIdx.x * upw;
    int64_t unit_end = (unit + upw < units_total) ? unit + upw : units_total;
    int seg_sliced = -1;
    if (segs > 0) {
        const int64_t j = blockIdx.x >> 3;
        const int64_t ubs = j % nub;
        seg_sliced = (int)((j / nub) * 8 + (blockIdx.x & 7));
        if (seg_sliced >= segs) return;
        unit = ubs * nst + (int64_t)seg_sliced * upw;
        unit_end = (unit + upw < (ubs + 1) * nst) ? unit + upw : (ubs + 1) * nst;
    }

    while (unit < unit_end) {
        const int64_t ub = unit / nst;
        const int64_t st0 = unit - ub * nst;
        const int64_t st1 = (st0 + (unit_end - unit) < nst) ? st0 + (unit_end - unit) : nst;
        const int seg = seg_sliced >= 0 ? seg_sliced : (int)((int64_t)blockIdx.x - (ub * nst) / upw);
        const int64_t user = ub * SC_USERS + ul;
        if (blockflag && blockflag[ub] == 0) {   // fallback pass: nobody in this user block asked for it (workgroup-uniform)
            unit += st1 - st0;
            continue;
        }

        float bq[X2 ? 1 : KH];
        float4 bqh[X2 ? NS16 : 1], bqm[X2 ? NS16 : 1];   // X2: 8 bf16 per step and plane, k = 16 s + 8 h + (0..7)
        if constexpr (X2) {
            const float4* qrow = reinterpret_cast<const float4*>(Q + user * D);   // [D hi | D mid] bf16 = D floats
            const bool uok = user < B;
#pragma unroll
            for (int s = 0; s < NS16; ++s) {
                bqh[s] = uok ? qrow[2 * s + h] : make_float4(0.f, 0.f, 0.f, 0.f);
                bqm[s] = uok ? qrow[D / 8 + 2 * s + h] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < NS16; ++s) {
                asm volatile("" : "+v"(bqh[s].x), "+v"(bqh[s].y), "+v"(bqh[s].z), "+v"(bqh[s].w));
                asm volatile("" : "+v"(bqm[s].x), "+v"(bqm[s].y), "+v"(bqm[s].z), "+v"(bqm[s].w));
            }
        } else {
            const float* qrow = Q + user * D;
            const bool uok = user < B;
#pragma unroll
            for (int s = 0; s < KH; ++s) bq[s] = uok ? qrow[2 * s + h] : 0.0f;
            // The query fragment is loop invariant.  Left as plain loads, the waitcnt bookkeeping merges "bq may still be in
            // flight" (first entry) with "the next stage's item prefetch is in flight" (back edge) at the stage loop's header
            // and puts s_waitcnt vmcnt(0) in front of the first MFMA of every stage.  Waiting once here and passing the
            // registers through an empty asm detaches them from the loads.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int s = 0; s < KH; ++s) asm volatile("" : "+v"(bq[s]));
        }
        double lk[KR];
#pragma unroll
        for (int j = 0; j < KR; ++j) lk[j] = KEMPTY;
        float thr = (user < B && dbg != 1 && (dbg < 5 || dbg == 7 || dbg == 8)) ? -INFINITY : INFINITY;
        float gbound = -INFINITY;   // the shared bound as last read
        unsigned genc = 0u;         // ... and the word in flight (split form: score_bound_k's estimate is there before the first stage)
        if (X2 && NB == 2 && gthr && user < B) genc = __hip_atomic_load(gthr + user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (NB == 3: polled in the loop)
        int qn = 0;
        bool quiet = false;   // the last half tile had no hit in any lane (wave-uniform)
        // LDS byte address of the queue tail (= qbase + 256 * qn) and the running item id of the next accumulator register
        const unsigned qbase = (unsigned)(size_t)(__attribute__((address_space(3))) float*)myqv;
        unsigned qaddr = qbase;
        int qid = (int)(st0 * SC_TI) + 4 * h;
        int64_t sc_cur = 0, sc_end = 0;
        int ns0 = NONE, ns1 = NONE;
        if (seen_ptr && user < B) {
            sc_cur = seen_ptr[user];
            sc_end = seen_ptr[user + 1];
            const int64_t first_item = st0 * SC_TI;
            int64_t lo = sc_cur, hi2 = sc_end;
            while (lo < hi2) {
                const int64_t mid = (lo + hi2) >> 1;
                if (seen_idx[mid] < first_item) lo = mid + 1; else hi2 = mid;
            }
            sc_cur = lo;
        }
        // The seen cursor keeps a window of the next two ids in registers.  It is refilled once per stage, BEFORE the item
        // prefetch is issued: the memory counter retires in order, so the wait for these two words in the first tile leaves
        // the prefetch in flight.  A lane that uses up its window inside one stage reloads on the spot (rare).
        const int NEED = -2;
        auto refill = [&]() {
            ns0 = sc_cur < sc_end ? (int)seen_idx[sc_cur] : NONE;
            ns1 = sc_cur + 1 < sc_end ? (int)seen_idx[sc_cur + 1] : NONE;
        };

        // fold every lane's queue into its register list: round e = entry e of every queue
        auto drain = [&]() {
            int rounds = 0;   // max over the wave of qn (< 32): bit by bit with ballots -- no cross-lane data movement
#pragma unroll
            for (int bit = 16; bit > 0; bit >>= 1)
                if (__ballot(qn >= (rounds | bit)) != 0ull) rounds |= bit;
            if (dbg == 2) rounds = 0;
            if (dbg == 3 || dbg == 7) {
                if (lane == 0) { atomicAdd(&g_sr_counters[0], 1ull); atomicAdd(&g_sr_counters[1], (unsigned long long)rounds); }
                atomicAdd(&g_sr_counters[2], (unsigned long long)qn);
            }
            if constexpr (X2) {
                // Split form: ONE list of 2 KR entries per user and segment, shared by the user's two lanes -- lane h = 0 holds ranks
                // 0 .. KR-1, lane h = 1 ranks KR .. 2KR-1 (56 f64 keys per lane instead of 112: the kernel stays out of scratch
                // memory, whose traffic had put a wait in front of every staged item tile).  A round folds in both lanes' queue
                // entries: the lower lane inserts its own entry and the partner's into its half; what falls off its end (o1, o2)
                // goes to the upper lane, which inserts it ONE ROUND LATER (same instruction stream for both lanes: "insert a,
                // insert b" with (a, b) = (own, partner's) below and (o1, o2 of the round before) above), plus one flush round.
                double o1 = KEMPTY, o2 = KEMPTY;
#pragma unroll 1
                for (int e = 0; e <= rounds; ++e) {
                    const int qid_e = e < QC ? myqi[e * 64] : -1;
                    const bool act = e < qn && (unsigned)qid_e < (unsigned)N;
                    const double k = act ? make_key(myqv[e * 64], qid_e) : KEMPTY;
                    const double kp = __shfl_xor(k, 32, 64), o1p = __shfl_xor(o1, 32, 64), o2p = __shfl_xor(o2, 32, 64);
                    const double a = h ? o1p : k, b = h ? o2p : kp;
                    double last = lk[KR - 1];
#pragma unroll
                    for (int j = KR - 1; j >= 1; --j) lk[j] = sr_max(lk[j], sr_min(a, lk[j - 1]));
                    lk[0] = sr_max(a, lk[0]);
                    o1 = sr_min(last, a);
                    last = lk[KR - 1];
#pragma unroll
                    for (int j = KR - 1; j >= 1; --j) lk[j] = sr_max(lk[j], sr_min(b, lk[j - 1]));
                    lk[0] = sr_max(b, lk[0]);
                    o2 = sr_min(last, b);
                }
            } else
#pragma unroll 1
            for (int e = 0; e < rounds; ++e) {   // ONE copy of the insertion code: the kernel must stay inside the I-cache
                const int qid_e = myqi[e * 64];
                const bool act = e < qn && (unsigned)qid_e < (unsigned)N;   // (not a catalog row: voided, or past the end)
                const double k = act ? make_key(myqv[e * 64], qid_e) : KEMPTY;
                // sorted insertion, best first: slot j takes k clamped into [lk[j], lk[j-1]]
#pragma unroll
                for (int j = KR - 1; j >= 1; --j) lk[j] = sr_max(lk[j], sr_min(k, lk[j - 1]));
                lk[0] = sr_max(k, lk[0]);
            }
            qn = 0;
            if (user < B && (dbg == 0 || dbg == 3 || dbg == 4 || dbg == 7 || dbg == 8)) {
                // The user's K-th best is at least (a) either lane's K-th best and (b) min(a, b) where a, b are the
                // two lanes' ceil(K/2)-th bests (K/2 items above a in one half + K/2 above b in the other): (b) is
                // close to the true K-th value because the halves are statistically alike -> ~40 % fewer hits.
                double kmid = lk[0], kkth = lk[0];
                if constexpr (KT > 0) {
                    // the m-th of both lanes is a bound on the 2m-th best of the pair: exact form m = ceil(K/2); split form (lists
                    // of KT >= K + 6 entries, and K + 6 is what the certificate wants the bound to speak of): m = KT/2
                    constexpr int MIDX = X2 ? KT / 2 : (KT + 1) / 2;
                    kmid = lk[MIDX - 1];
                    kkth = lk[KT - 1];
                } else {
#pragma unroll
                    for (int j = 1; j < KR; ++j) {
                        kmid = (j == (K + 1) / 2 - 1) ? lk[j] : kmid;
                        kkth = (j == K - 1) ? lk[j] : kkth;
                    }
                }
                const float mid = key_value(kmid), kth = key_value(kkth);
                const float pmid = __shfl_xor(mid, 32, 64), pkth = __shfl_xor(kth, 32, 64);
                if constexpr (X2) thr = h ? kth : pkth;   // the pair list's last entry (rank 2 KR, in the upper lane): a USER-level bound
                else thr = fmaxf(fmaxf(kth, pkth), fminf(mid, pmid));
                if (thr <= -3.402823466e+38f) thr = -INFINITY;   // lists not full yet
                if (dbg == 7) thr = -INFINITY;                   // (diagnostic: group bound only)
                // Every workgroup that scores items for this user holds such a lower bound of the user's K-th best: they share
                // the best one through a word in global memory (device-scope max on an order-preserving encoding; read back,
                // possibly stale, at the top of every stage).  Each segment then filters almost as if it had seen the whole
                // catalog: ~K(1+ln(N/K)) list insertions per USER instead of per (segment, lane).
                if (gthr) {
                    if (h == 0 && user < B && thr > -INFINITY) atomicMax(gthr + user, sr_enc(thr));
                    thr = fmaxf(thr, gbound);
                }
            }
        };

        // split form: stage st -> LDS buffer buf.  Instruction p of wave wid fills the 1 KB block (4 p + wid) of the buffer, lane l its
        // 16-byte slot l: rows are 4 D bytes (CPR = D/4 chunks), so that is chunk position l % CPR of row (4 p + wid) * (64 / CPR) +
        // l / CPR, which holds the row's chunk (l % CPR) ^ (row & 15)  (D = 64: 4 rows per block; D = 128: 2).
        // The lane's source address is  E + (64 st + srow_p) D + 4 g_p  with srow_p = (4 p + wid) (64 / CPR) + l / CPR: two base offsets
        // per lane (p even / odd: at D = 128 the swizzle term 8 p mod 16 alternates; at D = 64 they coincide) plus constants, so a
        // stage costs two 64-bit additions per instruction instead of a multiply-add and a row clamp each; only the catalog's last
        // stage (rows past N are clamped to the last row) goes the long way.
        constexpr int XCPR = D / 4, XRPB = 64 / XCPR;   // 16-byte chunks per row (16 / 32); rows per 1 KB block (4 / 2)
        [[maybe_unused]] const int xsrow0 = wid * XRPB + lane / XCPR, xsrow1 = (4 + wid) * XRPB + lane / XCPR;
        [[maybe_unused]] const int64_t xoff0 = (int64_t)xsrow0 * D + 4 * ((lane % XCPR) ^ (xsrow0 & 15));
        [[maybe_unused]] const int64_t xoff1 = (int64_t)xsrow1 * D + 4 * ((lane % XCPR) ^ (xsrow1 & 15));
        auto issue_stage = [&](int64_t st, int buf) {
            const bool last = (st + 1) * SC_TI > N;   // (wave-uniform)
            const float* sbase = E + st * (SC_TI * D);
#pragma unroll
            for (int p = 0; p < XCPR / 4; ++p) {
                const float* src = sbase + ((p & 1) ? xoff1 : xoff0) + (p >> 1) * (8 * XRPB * D);
                if (last) {
                    const int srow = (4 * p + wid) * XRPB + lane / XCPR;      // row of the stage this lane's slot belongs to
                    const int g = (lane % XCPR) ^ (srow & 15);               // ... and the row's chunk that lives in the slot
                    const int64_t row = st * SC_TI + srow;
                    src = E + (row < N ? row : N - 1) * D + 4 * g;
                }
                __attribute__((address_space(3))) unsigned char* dst =
                    (__attribute__((address_space(3))) unsigned char*)(__attribute__((address_space(3))) float*)tile + buf * (SC_TI * D * 4) + (4 * p + wid) * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, dst, 16, 0, 0);
            }
        };
        float4 pf[PF];
        auto prefetch = [&](int64_t st) {
            const int64_t item0 = st * SC_TI;
#pragma unroll
            for (int p = 0; p < PF; ++p) {   // always exactly PF loads (rows past N clamp to the last row; their scores are
                const int f = p * 256 + tid; // masked by index): a fixed count lets the waits in the tile loop leave them in flight
                const int64_t row = item0 + f / (D / 4);
                pf[p] = reinterpret_cast<const float4*>(E + (row < N ? row : N - 1) * D)[f % (D / 4)];
            }
        };
        if constexpr (X2) {
            refill();
            if constexpr (NB == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(ns0), "+v"(ns1) :: "memory");   // (no load pending into the loop)
            issue_stage(st0, 0);
            if constexpr (NB == 3) issue_stage(st0 + 1 < st1 ? st0 + 1 : st0, 1);   // (always: the waits below count on it)
        } else prefetch(st0);

        // Drains are WORKGROUP-wide where possible: the four waves share the stage barriers, so a wave draining alone stalls
        // the other three at the next barrier (measured: drain time x ~3).  The vote rides on the stage's first barrier (each
        // wave leaves a flag in LDS before it, everybody reads the four flags after it): a drain is called when some queue
        // holds more than SR_VOTE entries, which keeps SR_QC - 8 out of reach for the four half tiles of the stage; a wave whose
        // queues get there anyway (while the lists fill up at the start of a segment) drains on the spot at the top of a half tile.
#ifdef SC_PROFILE   // cycle accounting of one wave (make CXXFLAGS+=-DSC_PROFILE; scripts/tune_score.py), written out once at the end
        const bool prof = dbg >= 8 && blockIdx.x == 7 && wid == 0;
        long long ta = 0, tb = 0, tc = 0, td = 0, te = 0, t0 = 0, t1 = 0;
#define SC_T(x) x
#else
#define SC_T(x)
#endif
        for (int64_t st = st0; st < st1; ++st) {
            SC_T(if (prof) t0 = __builtin_readcyclecounter();)
            if (lane == 0) vote[wid] = 0;
            if (qn > vote_at) vote[wid] = 1;
            bool wg_drain;
            if constexpr (X2 && NB == 3) {
                if (((st - st0) & 15) == 0) {   // the bound word, every 16th stage (wave-uniform; this wait drains the stage in flight)
                    if (gthr && user < B) genc = __hip_atomic_load(gthr + user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(genc) :: "memory");   // (waited for HERE, not at the merge point below)
                }
                // this wave's part of stage st has landed; the stage behind it may still be in flight (the counter retires in order)
                if constexpr (D == 64) asm volatile("s_waitcnt vmcnt(4)" : "+v"(genc) :: "memory");
                else asm volatile("s_waitcnt vmcnt(8)" : "+v"(genc) :: "memory");
                // A bare barrier: __syncthreads() carries a fence, and the fence waits for EVERY outstanding load.  What has to be
                // ordered is done by hand: the wave's vote and its LDS reads of the previous stage are complete (lgkmcnt(0)); the vote
                // is read back in the same asm block (an LDS read the compiler knows about would wait for the loads in flight).
                f32x4 vt;
                asm volatile("s_waitcnt lgkmcnt(0)\n s_barrier\n ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)"
                             : "=v"(vt) : "v"((unsigned)(size_t)(__attribute__((address_space(3))) int*)vote) : "memory");
                wg_drain = (__float_as_uint(vt.x) | __float_as_uint(vt.y) | __float_as_uint(vt.z) | __float_as_uint(vt.w)) != 0u;
            } else {
                if constexpr (X2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the stage has landed in LDS
#ifndef SC_X_NOBARRIER
                __syncthreads();
#endif
                wg_drain = (vote[0] | vote[1] | vote[2] | vote[3]) != 0;
            }
            SC_T(if (prof) { t1 = __builtin_readcyclecounter(); ta += t1 - t0; t0 = t1; })
            [[maybe_unused]] const int buf = NB == 2 ? (int)(st - st0) & 1 : (int)((st - st0) % 3);
            if constexpr (!X2)
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const int f = p * 256 + tid;
                const int row = f / (D / 4);
                const int k0 = (f % (D / 4)) * 4;
                if constexpr (X2) {   // the split row as it is: [D bf16 hi | D bf16 mid], 16-byte pad per row
                    *reinterpret_cast<float4*>(tile + row * RSF + k0) = pf[p];
                } else {
                    float* dst = tile + row * RSF + (k0 >> 1);
                    *reinterpret_cast<float2*>(dst) = make_float2(pf[p].x, pf[p].z);
                    *reinterpret_cast<float2*>(dst + KH) = make_float2(pf[p].y, pf[p].w);
                }
            }
            if (wg_drain) {
                drain();
                qaddr = qbase;
            }
            SC_T(if (prof) { t1 = __builtin_readcyclecounter(); tb += t1 - t0; t0 = t1; })
#ifndef SC_X_NOBARRIER
            if constexpr (!X2) __syncthreads();   // (split form: one barrier per stage -- the other buffer is being filled, not this one)
#endif
            SC_T(if (prof) { t1 = __builtin_readcyclecounter(); tc += t1 - t0; t0 = t1; })
#ifndef SC_X_NOREFILL
            if constexpr (!X2) refill();
#endif
            // the shared bound: fold in the word requested one stage ago (no wait), request the next one -- before the item
            // prefetch, like the seen window, so that its wait leaves the prefetch in flight.
            // Split form: the seen window and the bound word of stage s + 1 are requested at the END of stage s (below) and are
            // first touched after the wait at the top of stage s + 1 -- the memory counter retires in order, and any wait for
            // them in the middle of a stage would also wait for the stage loads issued just after them.
            if (gthr) {
                gbound = fmaxf(gbound, sr_dec(genc));
                thr = fmaxf(thr, gbound);
                if constexpr (!X2) {
                    if (user < B) genc = __hip_atomic_load(gthr + user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#ifndef SC_X_NOPREFETCH
            if constexpr (X2) {
                if constexpr (NB == 2) { if (st + 1 < st1) issue_stage(st + 1, buf ^ 1); }
                else issue_stage(st + 2 < st1 ? st + 2 : st1 - 1, (buf + 2) % 3);   // every wave is past stage st - 1, the buffer's last reader
            }
            else prefetch(st + 1 < st1 ? st + 1 : st);
#endif
            // The tile loop, in HALF tiles (16 items = accumulator registers 0-7 / 8-15): one queue-room check -- and the only
            // in-loop copy of the drain code -- serves both halves; the MFMA chain runs in the even iterations.
            f32x16 acc;
#pragma unroll 1
            for (int ht = 0; ht < 2 * (SC_TI / 32); ++ht) {
                const int64_t item0 = st * SC_TI + (ht >> 1) * 32;   // first item of the tile
                if (item0 >= N) break;
                // a half tile appends up to 8 entries per lane: make room now (while the lists fill up, or after a burst)
                if (__ballot(qn > QC - 8) != 0ull) {
                    drain();
                    qaddr = qbase;
                }
                const int hend = (int)item0 + 16 + 16 * (ht & 1);    // end of this half's item range
                // Filter + append.  fp32 MFMA and vector instructions share one pipe on gfx950 (scripts/micro/mfma_valu_samewave.hip:
                // every vector instruction costs its ~5-6 cycles on top of the MFMA chain, from the same wave or another), so the
                // vector instruction count per score IS the kernel's efficiency: 4 here (SR_APPEND1), no hit mask, no exec juggling.
                // ">=" and not ">": thr may come from OTHER items (the partner lane, other workgroups), and an item that ties
                // the bound with a lower id can still belong to the top K; the f64 keys order whatever gets through.  (Neither a
                // score -- an fmaf chain started at +0 -- nor a bound is ever -0.)  What must never get in -- the user's seen
                // items, rows past the end of the catalog -- is voided AFTER the fact: a lane whose seen cursor points into
                // these 16 items looks for that id among the entries it has just appended (usually none or one) and negates it;
                // the drain skips entries whose id is not a catalog row.
#define SR_HALF(PRE, A0, A1, A2, A3, A4, A5, A6, A7)                                                                             \
    asm volatile(PRE SR_APPEND1("%5", "", 1) SR_APPEND1("%6", "", 1) SR_APPEND1("%7", "", 1) SR_APPEND1("%8", "", 5)              \
                     SR_APPEND1("%9", "", 1) SR_APPEND1("%10", "", 1) SR_APPEND1("%11", "", 1) SR_APPEND1("%12", "", 5)           \
                 : "+v"(qn), "+v"(qaddr), "+v"(qid)                                                                             \
                 : "v"(thr), "v"(qbase), "v"(A0), "v"(A1), "v"(A2), "v"(A3), "v"(A4), "v"(A5), "v"(A6), "v"(A7),                 \
                   [qoff] "i"(4 * QC)                                                                                           \
                 : "vcc", "memory")
                const int qn0 = qn;
                auto void_seen = [&]() {   // seen cursor: two ids prefetched per stage, one id per lane per pass
                    while (__ballot(ns0 < hend) != 0ull) {
                        if (ns0 < hend) {
                            if (ns0 == NEED) {   // window used up inside this stage: reload and wait here, inside the rare branch
                                refill();
                                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                                asm volatile("" : "+v"(ns0), "+v"(ns1));
                            } else {
                                if (ns0 >= hend - 16 && ((ns0 >> 2) & 1) == h)   // (bit 2 of the item id: which lane of the pair)
                                    for (int e = qn0; e < qn; ++e)
                                        if (myqi[e * 64] == ns0) myqi[e * 64] = -1;
                                ++sc_cur;
                                ns0 = ns1;
                                ns1 = NEED;
                            }
                        }
                    }
                };
                if ((ht & 1) == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                    const float* arow = tile + ((ht >> 1) * 32 + c) * RSF + h * KH;
#ifndef SC_X_NOMFMA
                    if constexpr (X2) {
                        // lane (c, h) holds item row c, k = 16 s + 8 h + (0..7): 16 bytes at float offset 8 s + 4 h of a plane
                        // (rows are 4 D + 16 bytes apart: the 16 lanes of a ds_read_b128 pass hit 16 different 16-byte bank groups)
                        // row (ht >> 1) * 32 + c of buffer buf; chunk j of a row sits at position j ^ (row & 15) = j ^ (c & 15)
                        const char* xbase = reinterpret_cast<const char*>(tile) + buf * (SC_TI * D * 4) + ((ht >> 1) * 32 + c) * (D * 4);
                        const int c15 = c & 15;
                        // Two passes over one set of fragment registers: first the hi plane (products hi.hi and hi.mid), then the mid
                        // plane (mid.hi).  All NS16 reads of a pass are in flight together (one wait; left alone, the register
                        // allocator funnels them through one register quad and exposes an LDS round trip in front of every other
                        // MFMA), and NS16 instead of 2 NS16 quads are live beside the lists.
                        // The reads are inline asm: the compiler orders every LDS read it knows about behind ALL outstanding
                        // global->LDS loads (s_waitcnt vmcnt(0) -- it cannot see that the next stage's loads fill the OTHER buffer),
                        // which would put the whole load latency in front of every tile.
                        f32x4 af[4];
                        const unsigned xb = (unsigned)(size_t)(__attribute__((address_space(3))) const char*)xbase;
#define SX_FRAGS(J0)                                                                                                              \
    asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %5\n ds_read_b128 %2, %6\n ds_read_b128 %3, %7\n s_waitcnt lgkmcnt(0)"      \
                 : "=&v"(af[0]), "=&v"(af[1]), "=&v"(af[2]), "=&v"(af[3])                                                          \
                 : "v"(xb + ((((J0) + 0 + h) ^ c15) << 4)), "v"(xb + ((((J0) + 2 + h) ^ c15) << 4)),                               \
                   "v"(xb + ((((J0) + 4 + h) ^ c15) << 4)), "v"(xb + ((((J0) + 6 + h) ^ c15) << 4))                                \
                 : "memory")
                        // hi plane = chunks 2 s + h (s < NS16), mid plane = chunks D/8 + 2 s + h; four steps (64 k) per pass
#pragma unroll
                        for (int g4 = 0; g4 < NS16 / 4; ++g4) {
                            SX_FRAGS(8 * g4);
#pragma unroll
                            for (int s = 0; s < 4; ++s) {
                                const bf16x8 xh = __builtin_bit_cast(bf16x8, af[s]);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, __builtin_bit_cast(bf16x8, bqm[4 * g4 + s]), acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, __builtin_bit_cast(bf16x8, bqh[4 * g4 + s]), acc, 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int g4 = 0; g4 < NS16 / 4; ++g4) {
                            SX_FRAGS(D / 8 + 8 * g4);
#pragma unroll
                            for (int s = 0; s < 4; ++s)
                                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[s]), __builtin_bit_cast(bf16x8, bqh[4 * g4 + s]), acc, 0, 0, 0);
                        }
#undef SX_FRAGS
                    } else {
#pragma unroll
                    for (int q = 0; q < KH / 4; ++q) {
#ifdef SC_X_NOAFRAG
                        float4 a = make_float4(bq[0], bq[1], bq[2], bq[3]); asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w));
#else
                        const float4 a = *reinterpret_cast<const float4*>(arow + 4 * q);
#endif
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[4 * q + 0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[4 * q + 1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[4 * q + 2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[4 * q + 3], acc, 0, 0, 0);
                    }
                    }
#endif
                    SC_T(if (prof) { asm volatile("s_nop 0" :: "v"(acc[0])); t1 = __builtin_readcyclecounter(); td += t1 - t0; t0 = t1; })
                    // The 18 wait states an fp32 32x32 MFMA result needs before a vector instruction touches it: the compiler
                    // cannot see into the asm blocks below, so they are spent here, once, on every path (tied to acc so that
                    // the chain cannot sink below them; scripts/lint_mfma_hazard.py checks the final ISA).
                    asm volatile("s_nop 15\n s_nop 1" : "+v"(acc));
#ifndef SC_X_NOAPPEND
                    // Long catalogs: once the bounds are warm, most half tiles hold no hit for any of the wave's 64 lanes.  After a
                    // half tile without hits the next one is screened first -- a v_max3 tree over its 8 scores and one compare,
                    // 5 instructions instead of 40 -- and skipped if nothing can pass.  (Beauty-sized catalogs never get quiet.)
                    bool run = true;
                    if (quiet) {
                        float mx;   // (raw v_max3: fmaxf would re-canonicalise every operand)
                        asm volatile("v_max3_f32 %0, %1, %2, %3\n v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %0, %0, %6, %7\n v_max_f32 %0, %0, %8"
                                     : "=&v"(mx) : "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]));
                        run = __ballot(mx >= thr) != 0ull;
                    }
                    if (run) {
                        SR_HALF("", acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]);
                        quiet = __ballot(qn != qn0) == 0ull;
                    } else {
                        qid += 16;
                    }
#endif
                } else {
#ifndef SC_X_NOAPPEND
                    bool run = true;
                    if (quiet) {
                        float mx;
                        asm volatile("v_max3_f32 %0, %1, %2, %3\n v_max3_f32 %0, %0, %4, %5\n v_max3_f32 %0, %0, %6, %7\n v_max_f32 %0, %0, %8"
                                     : "=&v"(mx) : "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15]));
                        run = __ballot(mx >= thr) != 0ull;
                    }
                    if (run) {
                        SR_HALF("", acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15]);
                        quiet = __ballot(qn != qn0) == 0ull;
                    } else {
                        qid += 16;
                    }
#endif
                }
                void_seen();
#undef SR_HALF
                SC_T(if (prof) { t1 = __builtin_readcyclecounter(); te += t1 - t0; t0 = t1; })
            }
            if constexpr (X2 && NB == 2) {   // requests for the next stage (see above)
                refill();
                if (gthr && user < B) genc = __hip_atomic_load(gthr + user, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
#ifdef SC_PROFILE
        if (prof && lane == 0) {
            g_sr_counters[0] = (unsigned long long)ta;   // wait at the stage's first barrier
            g_sr_counters[1] = (unsigned long long)tb;   // LDS stores of the staged items (+ workgroup drain)
            g_sr_counters[2] = (unsigned long long)tc;   // wait at the second barrier
            g_sr_counters[3] = (unsigned long long)td;   // prefetch issue + A-fragment reads + MFMA chain (both tiles)
            g_sr_counters_x[0] = (unsigned long long)te; // filter, seen-mask, appends (+ wave-local drains)
            g_sr_counters_x[1] = (unsigned long long)(st1 - st0);
        }
#endif
        drain();
        // Split form: the lists are SHORTER than the K the caller wants (capacity K here = c, see score_topk_impl), so what a lane
        // has dropped matters: everything it filtered out was below its threshold of the moment, everything its list pushed out
        // is below the list's last entry -- both <= the final threshold, which goes out with the list (score_topk_merge_x).
        if (part_T && user < B && h == 0) part_T[user * maxseg + seg] = thr;
        // The two partial lists of every user (K values + K ids per lane).  Written lane by lane this is 2K scattered 4-byte
        // stores per lane (64 cache lines per store instruction: 11 % of the kernel on the Beauty shape, scripts ablation
        // SC_X_NOOUTPUT); instead the wave transposes them through its (now empty) queue memory and writes each user's
        // 2K contiguous floats with 16-byte stores.
#ifdef SC_X_NOOUTPUT
        if (lk[0] == 12345.0) {
#else
        {
#endif
            const int K2 = 2 * K;
            float* ldsA = qv + wid * QC * 64;                                  // 2 x QC*64 floats of wave-private LDS
            float* ldsB = reinterpret_cast<float*>(qi + wid * QC * 64);
            const int64_t user0 = ub * SC_USERS + wid * 32;                        // the wave's first user
            const int64_t ustride = (int64_t)maxseg * K2;                          // floats between consecutive users
            const int64_t gbase = (user0 * maxseg + seg) * K2;
            if ((K & 1) == 0 && 64 * K <= 2 * QC * 64) {
#pragma unroll 1
                for (int pass = 0; pass < 2; ++pass) {
                    const int f0 = c * K2 + h * K;
#pragma unroll
                    for (int j = 0; j < KR; ++j)
                        if (j < K) {
                            const bool empty = lk[j] == KEMPTY;
                            const int f = f0 + j;
                            float* slot = f < QC * 64 ? ldsA + f : ldsB + (f - QC * 64);
                            if (pass == 0) *slot = empty ? -INFINITY : key_value(lk[j]);
                            else *reinterpret_cast<int*>(slot) = empty ? -1 : key_item(lk[j]);
                        }
                    float* gdst = (pass == 0 ? part_vals : reinterpret_cast<float*>(part_idx)) + gbase;
                    for (int i = lane; i < 16 * K; i += 64) {      // 32 users x 2K floats = 16K float4
                        const int f = 4 * i;
                        const int cc = f / K2, e = f - cc * K2;
                        const float* slot = f < QC * 64 ? ldsA + f : ldsB + (f - QC * 64);
                        if (user0 + cc < B) *reinterpret_cast<float4*>(gdst + cc * ustride + e) = *reinterpret_cast<const float4*>(slot);
                    }
                }
            } else if (user < B) {
                float* pv = part_vals + ((user * maxseg + seg) * 2 + h) * K;
                int* pi = part_idx + ((user * maxseg + seg) * 2 + h) * K;
#pragma unroll
                for (int j = 0; j < KR; ++j)
                    if (j < K) {
                        const bool empty = lk[j] == KEMPTY;
                        pv[j] = empty ? -INFINITY : key_value(lk[j]);
                        pi[j] = empty ? -1 : key_item(lk[j]);
                    }
            }
        }
        unit += st1 - st0;
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// merge the per-segment partial lists of one user (one wave per user), sort, apply the K > #unmasked fill.
__device__ __forceinline__ void bitonic_step(float& v, int& i, int j, bool keep_first, int lane) {
    const float pv = __shfl_xor(v, j, 64);
    const int pi = __shfl_xor(i, j, 64);
    const bool take = keep_first ? sc_before(pv, pi, v, i) : sc_before(v, i, pv, pi);
    if (take) { v = pv; i = pi; }
    (void)lane;
}

__device__ __forceinline__ void bitonic_sort64(float& v, int& i, int lane) {
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            const bool lower = (lane & j) == 0;
            const bool asc = (lane & k) == 0;
            bitonic_step(v, i, j, lower == asc, lane);
        }
}

// write one user's sorted list (lane j = j-th best; PAD = no entry) and apply the K > #unmasked fill
__device__ __forceinline__ void topk_emit(float bv, int bi, int lane, int64_t user, int64_t N, int K,
                                          const int64_t* __restrict__ seen_ptr, const int64_t* __restrict__ seen_idx,
                                          float* __restrict__ vals, int64_t* __restrict__ idx) {
    const int PAD = 0x7FFFFFFF;
    const int nvalid = __popcll(__ballot(bi != PAD && lane < K));
    if (lane < K && bi != PAD) {
        vals[user * K + lane] = bv;
        idx[user * K + lane] = bi;
    }
    if (nvalid < K && lane == 0) {
        // fewer than K unmasked items: torch.topk would continue into the masked (-1e23) entries;
        // ties -> lowest index, i.e. the user's seen items in ascending order.
        int o = nvalid;
        int64_t last = -1;
        if (seen_ptr)
            for (int64_t p = seen_ptr[user]; p < seen_ptr[user + 1] && o < K; ++p) {
                const int64_t it = seen_idx[p];
                if (it < 0 || it >= N || it == last) continue;
                last = it;
                vals[user * K + o] = RE_MASKED_SCORE;
                idx[user * K + o] = it;
                ++o;
            }
        for (; o < K; ++o) { vals[user * K + o] = -INFINITY; idx[user * K + o] = -1; }
    }
}

__global__ __launch_bounds__(256) void score_topk_merge(const float* __restrict__ part_vals, const int* __restrict__ part_idx,
                                                        int maxseg, int lps, int presorted, int64_t B, int64_t N, int K, int64_t nst, int64_t upw,
                                                        const int64_t* __restrict__ seen_ptr, const int64_t* __restrict__ seen_idx,
                                                        float* __restrict__ vals, int64_t* __restrict__ idx,
                                                        const int* __restrict__ userflag, int segs) {
    const int lane = threadIdx.x & 63;
    const int64_t user = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (user >= B) return;
    if (userflag && userflag[user] == 0) return;   // fallback pass: this user's result is certified already
    const int64_t ub = user / SC_USERS;
    const int64_t w0 = (ub * nst) / upw, w1 = ((ub + 1) * nst - 1) / upw;
    const int nseg = (segs > 0 ? segs : (int)(w1 - w0 + 1)) * lps;   // lps partial lists per segment (2 for the register-list variant)
    maxseg *= lps;
    const int PAD = 0x7FFFFFFF;
    float bv = -INFINITY;
    int bi = PAD;
    for (int s = 0; s < nseg; ++s) {
        float v = -INFINITY;
        int i = PAD;
        if (lane < K) {
            v = part_vals[(user * maxseg + s) * K + lane];
            i = part_idx[(user * maxseg + s) * K + lane];
            if (i < 0) { i = PAD; v = -INFINITY; }
        }
        if (!presorted) bitonic_sort64(v, i, lane);   // register-list partials arrive sorted best-first (padding last)
        // top-64 of the union of two best-first lists: compare lane t with lane 63-t of the other list
        const float rv = __shfl(v, 63 - lane, 64);
        const int ri = __shfl(i, 63 - lane, 64);
        if (sc_before(rv, ri, bv, bi)) { bv = rv; bi = ri; }
#pragma unroll
        for (int j = 32; j > 0; j >>= 1) bitonic_step(bv, bi, j, (lane & j) == 0, lane);
    }
    topk_emit(bv, bi, lane, user, N, K, seen_ptr, seen_idx, vals, idx);
}

// ---------------------------------------------------------------------------------------------------------
// The split form (score_kernel_reg<.., X2 = true>): preparation, candidate re-scoring, certificate.
//
// score_split_k: fp32 rows -> two bf16 planes by round-to-nearest-even, hi = bf16(x), mid = bf16(x - hi) (x - hi is exact in
// fp32), so x = hi + mid + r with |r| <= 2^-18 |x|.  Output row = [D hi | D mid] (4 D bytes, the fp32 row's size); also the
// row's Euclidean norm (rounded up) -> rownorm[] and/or a device-wide maximum (one atomicMax per wave at the end).
__device__ __forceinline__ unsigned sx_bf16(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
template <int D>
__global__ __launch_bounds__(256) void score_split_k(const float* __restrict__ X, int64_t R, unsigned short* __restrict__ Xs,
                                                     float* __restrict__ rownorm, unsigned* __restrict__ maxnorm,
                                                     uint4* __restrict__ zfill, size_t zfill_n16) {
    constexpr int LPR = D / 4;   // lanes per row (16 or 32: a wave holds whole rows)
    const int64_t total = R * LPR;
    // side job of the query split: zero the call's flag / bound words (one launch less than a zero-fill kernel of its own)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < zfill_n16; i += (size_t)gridDim.x * 256) zfill[i] = make_uint4(0u, 0u, 0u, 0u);
    float wmax = 0.0f;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < total; base += (int64_t)gridDim.x * 256) {
        const int64_t f = base + threadIdx.x;
        const bool ok = f < total;
        const int64_t row = ok ? f / LPR : 0;
        const int kq = (int)(f % LPR);
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) x = reinterpret_cast<const float4*>(X)[f];
        const float xv[4] = {x.x, x.y, x.z, x.w};
        unsigned hi[4], mid[4];
        double ss = 0.0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            hi[j] = sx_bf16(xv[j]);
            mid[j] = sx_bf16(xv[j] - __uint_as_float(hi[j] << 16));
            ss += (double)xv[j] * (double)xv[j];
        }
        if (ok) {
            unsigned short* dst = Xs + row * (2 * D) + 4 * kq;
            *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
            *reinterpret_cast<uint2*>(dst + D) = make_uint2(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16));
        }
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
        float nrm = ss > 1e-30 ? sqrtf((float)ss) : (float)sqrt(ss);   // (fp64 sqrt only for rows that would underflow in fp32)
        nrm = nrm * 1.000001f;                         // rounded up (sqrt, conversion)
        if (ss > 0.0 && nrm < 1.2e-38f) nrm = 1.2e-38f;  // never 0 for a non-zero row
        if (ok && kq == 0 && rownorm) rownorm[row] = nrm;
        if (ok) wmax = (nrm > wmax || nrm != nrm) ? nrm : wmax;   // (a NaN row poisons the maximum: every user falls back)
    }
    if (maxnorm) {
        unsigned e = __float_as_uint(wmax);   // non-negative floats (and NaN above them) order as unsigned words
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e = max(e, (unsigned)__shfl_xor((int)e, o, 64));
        __shared__ unsigned wm[4];
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) {   // one atomic per workgroup: same-address atomics serialise at ~10 ns each
            e = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
            if (e != 0u) atomicMax(maxnorm, e);
        }
    }
}

// score_bound_k: a starting threshold for every user from a strided SAMPLE of the catalog (n = 32 n_tiles items, every
// stride-th row of the split table), before the main kernel runs.  Nearly all list insertions of the main kernel are warm-up:
// every (segment, lane) list starts empty and lets through ~K ln(n/K) items before its own bound is any good (measured on
// Beauty: ~1300 insertions per user for 56 final entries).  The r-th best score of a sample that holds a fraction f of the
// catalog sits near catalog rank r/f; with r = (K + 6) f + 4.5 sqrt((K + 6) f) + 2 fewer than 1 user in 10^4 has r or more
// of its best K + 6 items inside the sample, i.e. a bound above its (K + 6)-th best score.  The bound need not be valid: it
// is a filter threshold like any other, it is part of the dropped-below bound T the lists report, and a user for whom it was
// too high fails the certificate in score_topk_merge_x and is redone exactly.  (Only the split form may do this.)
// One workgroup = 32 users, two waves (even / odd sample tiles, lists joined at the end); per tile 3 D/16 MFMAs and 16 sorted insertions into the lane's best-8 list (v_med3 per slot); the
// bound is the smaller of the pair's two rhalf-th bests (2 rhalf items of the sample are at least that good).
#ifndef SC_BOUND_SIGMAS
#define SC_BOUND_SIGMAS 4.5
#endif
template <int D>
__global__ __launch_bounds__(128) void score_bound_k(const float* __restrict__ Qs, const float* __restrict__ Es, int64_t B,
                                                     int n_tiles, int64_t stride, int rhalf, unsigned* __restrict__ gthr,
                                                     int chunk_tiles, float* __restrict__ partial) {
    constexpr int NS16 = D / 16;
    __shared__ float lx[8 * 64];
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6;
    const int64_t user = (int64_t)blockIdx.x * 32 + c;   // 32 users per workgroup, two waves: even / odd sample tiles
    float4 bqh[NS16], bqm[NS16];
    {
        const float4* qrow = reinterpret_cast<const float4*>(Qs + user * D);
        const bool uok = user < B;
#pragma unroll
        for (int s = 0; s < NS16; ++s) {
            bqh[s] = uok ? qrow[2 * s + h] : make_float4(0.f, 0.f, 0.f, 0.f);
            bqm[s] = uok ? qrow[D / 8 + 2 * s + h] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = -INFINITY;
    float4 xh[NS16], xm[NS16], nh[NS16], nm[NS16];
    auto fetch = [&](int t, float4* fh, float4* fm) {   // the lane's A row of tile t: item (32 t + c) * stride
        const float4* xr = reinterpret_cast<const float4*>(Es + ((int64_t)(t * 32 + c) * stride) * D);
#pragma unroll
        for (int s = 0; s < NS16; ++s) { fh[s] = xr[2 * s + h]; fm[s] = xr[D / 8 + 2 * s + h]; }
    };
    auto insert = [&](float v) {
#pragma unroll
        for (int j = 7; j >= 1; --j) l[j] = __builtin_amdgcn_fmed3f(v, l[j], l[j - 1]);   // clamp(v, l[j], l[j-1]): sorted insertion
        l[0] = fmaxf(v, l[0]);
    };
    // few users against a long catalog: the sample is cut into gridDim.y chunks of chunk_tiles tiles (one workgroup each, so that
    // the sample can be large -- N/32 items -- without a few workgroups walking all of it); score_bound_merge_k joins the chunks
    const int t_beg = blockIdx.y * chunk_tiles;
    const int t_end = t_beg + chunk_tiles < n_tiles ? t_beg + chunk_tiles : n_tiles;
    if (t_beg + wv < t_end) fetch(t_beg + wv, xh, xm);
    for (int t = t_beg + wv; t < t_end; t += 2) {
        fetch(t + 2 < t_end ? t + 2 : t, nh, nm);   // (in flight under this tile's MFMAs and insertions)
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < NS16; ++s) {
            const bf16x8 ah = __builtin_bit_cast(bf16x8, xh[s]), am = __builtin_bit_cast(bf16x8, xm[s]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, __builtin_bit_cast(bf16x8, bqh[s]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bqm[s]), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(bf16x8, bqh[s]), acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) insert(acc[r]);
#pragma unroll
        for (int s = 0; s < NS16; ++s) { xh[s] = nh[s]; xm[s] = nm[s]; }
    }
    // the odd wave hands its lists over; a list that lost entries beyond its 8 only makes the bound lower (safe side)
    if (wv == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) lx[j * 64 + lane] = l[j];
    }
    __syncthreads();
    if (wv == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) insert(lx[j * 64 + lane]);
        if (partial) {   // one of several chunks: the lane's best 8 of this chunk go to score_bound_merge_k
#pragma unroll
            for (int j = 0; j < 8; ++j) partial[(((int64_t)blockIdx.x * gridDim.y + blockIdx.y) * 8 + j) * 64 + lane] = l[j];
            return;
        }
        float mine = l[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) mine = (j == rhalf - 1) ? l[j] : mine;
        const float bound = fminf(mine, __shfl_xor(mine, 32, 64));
        if (h == 0 && user < B && bound > -INFINITY) gthr[user] = sr_enc(bound);
    }
}

// score_front_k: the three launches in front of the split form's main kernel -- query split (+ zeroing of the call's words), item-table
// split, starting thresholds -- as ONE launch of independent workgroups (round 5; 7 + 7 + 21 us in three launches before):
//   blocks [nbe, nbe + ugroups): score_bound_k's job for 64 users, on the fp32 rows: the users' query rows are split here (score_split_k's arithmetic, bit
//     for bit: the planes go to Qs for the main kernel, the norms to qnorm) and the sample's item rows are rounded to bf16 on the fly
//     (v_cvt_pk_bf16_f32; a threshold need not be valid, only reported honestly) -- so the job waits for no other launch; four
//     waves take the sample tiles t = wave, wave + 4, ... and wave 0 joins the four best-8 lists; the bound word is WRITTEN (0 = none), so it
//     needs no zeroing;
//   blocks [0, nbe): score_split_k's job on the item table, the device-wide maximum norm as one word PER BLOCK (bmax[j]: no
//     atomic, nothing to zero; score_topk_merge_x takes the maximum of the nbe words);
//   every block: a share of the words that must be zero when the main kernel starts (second thresholds, user / block flags) -- none of them is
//     written inside this launch.
#ifdef RE_DEBUG
__device__ unsigned long long g_front_stamps[256 * 16];   // [block < 256][16]: wave 0's s_memtime at the phases of score_front_k (scripts/front_stamps.py)
extern "C" int re_dbg_front_stamps(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_front_stamps), sizeof(g_front_stamps)) == hipSuccess ? 0 : 1; }
__device__ unsigned long long g_front_wall[1024 * 2];     // [block < 1024][start, end]: wall_clock64 (100 MHz, one clock for the whole device)
extern "C" int re_dbg_front_wall(unsigned long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_front_wall), sizeof(g_front_wall)) == hipSuccess ? 0 : 1; }
#define FRONT_STAMP(I) do { if (threadIdx.x == 0 && blockIdx.x < 256) g_front_stamps[blockIdx.x * 16 + (I)] = __builtin_readcyclecounter(); \
                            if (threadIdx.x == 0 && blockIdx.x < 1024 && ((I) == 0 || (I) == 11)) g_front_wall[blockIdx.x * 2 + ((I) == 11)] = wall_clock64(); } while (0)   // (blocks 0 .. nbe - 1 are the splitting ones)
#else
#define FRONT_STAMP(I) do { } while (0)
#endif
#define SF_NW 8          // waves of a score_front_k workgroup: the tile phases of its 64 users (a wave takes sample tiles wave, wave + SF_NW, ...)
template <int D>
__global__ __launch_bounds__(64 * SF_NW) void score_front_k(const float* __restrict__ Q, const float* __restrict__ E, int64_t B, int64_t N,
                                                     unsigned short* __restrict__ Qs, float* __restrict__ qnorm, unsigned short* __restrict__ Es,
                                                     unsigned* __restrict__ bmax, int ugroups, int nbe, int n_tiles, int64_t stride, int rhalf,
                                                     unsigned* __restrict__ gthr, unsigned* __restrict__ zero_a, size_t zero_a_n,
                                                     unsigned* __restrict__ zero_b, size_t zero_b_n) {
    constexpr int NS16 = D / 16;
    __shared__ __align__(16) float lx2[(64 * (D + 4) > SF_NW * 2 * 8 * 64) ? 64 * (D + 4) : SF_NW * 2 * 8 * 64];   // the 64 users' query rows, then the waves' lists
    __shared__ unsigned wm[SF_NW];
    __shared__ uint4 bq[2 * (D / 16) * 64];
    __shared__ __align__(16) unsigned atile[SF_NW * 32 * (D / 2 + 4)];
    FRONT_STAMP(0);
    {
        const size_t nb = (size_t)gridDim.x * (64 * SF_NW), i0 = (size_t)blockIdx.x * (64 * SF_NW) + threadIdx.x;
        for (size_t i = i0; i < zero_a_n; i += nb) zero_a[i] = 0u;
        for (size_t i = i0; i < zero_b_n; i += nb) zero_b[i] = 0u;
    }
    if ((int)blockIdx.x < nbe) {                     // (the splitting workgroups come FIRST in the grid: theirs is the longest chain of memory round trips)
        // ---- item-table split (score_split_k's loop; rows of this block: a grid-stride share)
        constexpr int LPR = D / 4;
        const int64_t total = N * LPR;
        float wmax = 0.0f;
        // (four rounds' loads in flight at once: a round is one trip to HBM, ~2 us, and a block has up to 16 of them)
        constexpr int U = 4;
        constexpr int NT = 64 * SF_NW;
        for (int64_t base = (int64_t)blockIdx.x * NT; base < total; base += (int64_t)nbe * NT * U) {
            float4 xs[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t f = base + (int64_t)u * nbe * NT + threadIdx.x;
                xs[u] = f < total ? reinterpret_cast<const float4*>(E)[f] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t f = base + (int64_t)u * nbe * NT + threadIdx.x;
                const bool ok = f < total;
                const int64_t row = ok ? f / LPR : 0;
                const int kq = (int)(f % LPR);
                const float xv[4] = {xs[u].x, xs[u].y, xs[u].z, xs[u].w};
                unsigned hi[4], mid[4];
                double ss = 0.0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    hi[j] = sx_bf16(xv[j]);
                    mid[j] = sx_bf16(xv[j] - __uint_as_float(hi[j] << 16));
                    ss += (double)xv[j] * (double)xv[j];
                }
                if (ok) {
                    unsigned short* dst = Es + row * (2 * D) + 4 * kq;
                    *reinterpret_cast<uint2*>(dst) = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
                    *reinterpret_cast<uint2*>(dst + D) = make_uint2(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16));
                }
#pragma unroll
                for (int o = LPR / 2; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
                float nrm = ss > 1e-30 ? sqrtf((float)ss) : (float)sqrt(ss);
                nrm = nrm * 1.000001f;
                if (ss > 0.0 && nrm < 1.2e-38f) nrm = 1.2e-38f;
                if (ok) wmax = (nrm > wmax || nrm != nrm) ? nrm : wmax;
            }
        }
        unsigned e = __float_as_uint(wmax);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) e = max(e, (unsigned)__shfl_xor((int)e, o, 64));
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = e;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned em = wm[0];
#pragma unroll
            for (int w = 1; w < SF_NW; ++w) em = max(em, wm[w]);
            bmax[blockIdx.x] = em;
        }
        FRONT_STAMP(11);
        return;
    }
    // ---- starting thresholds of 64 users (score_bound_k's job, two groups of 32 users per wave)
    // What a workgroup's time goes into is the L1's line rate: a wave-load whose lanes each take 16 bytes of their own row touches 32 cache
    // lines (32+ cycles in the texture path, x 8 loads a tile x 12 waves a CU = the 3 700 cycles a tile measured with one group per wave).
    // So: every item tile a wave loads serves TWO groups of users (same A fragments, two B operands), and the 64 users' query rows come in
    // coalesced once per workgroup through LDS (16-byte pieces, consecutive lanes: 8 lines an instruction) instead of row-per-lane by every wave.
    const int lane = threadIdx.x & 63, c = lane & 31, h = lane >> 5, wv = threadIdx.x >> 6, gb = (int)blockIdx.x - nbe;   // gb: the workgroup's 64 users
    constexpr int QLD = D + 4;                                   // LDS row stride (floats): + 16 bytes, so a row-per-lane b128 read is conflict-free
    float* const qs = lx2;                                       // [64 users][QLD]
    // An item tile (32 sample rows) comes in COALESCED -- lane L takes 16 bytes at offset 16 L of a 1 KB piece (4 rows at D = 64): 8 cache lines an
    // instruction, where a row per lane touched 32 -- is rounded to bf16 and goes through the wave's own LDS buffer ([row][D bf16 + 16 B]) into
    // the MFMA's A layout (lane (c, h): 8 k of row c).  The wave writes and reads its own buffer: no barrier.
    constexpr int F4 = D / 4, RPI = 64 / F4, NLD = 32 / RPI;     // float4 per row; rows per load instruction; load instructions per tile
    constexpr int ARS = D / 2 + 4;                               // the buffer's row stride in dwords (16-byte aligned rows)
    unsigned* const at = atile + wv * 32 * ARS;
    const int lrow = lane / F4, lk4 = lane % F4;
    float4 raw[NLD];
    const int64_t jstep = (int64_t)RPI * stride * D;             // floats between the rows of two consecutive load instructions (uniform)
    auto fetch = [&](int t) {
        const float* base = E + ((int64_t)(t * 32 + lrow) * stride) * D + 4 * lk4;   // sample row r = item r stride
#pragma unroll
        for (int j = 0; j < NLD; ++j) raw[j] = *reinterpret_cast<const float4*>(base + j * jstep);
    };
    if (wv < n_tiles) fetch(wv);                                 // (the first tile's rows: requested before the query rows, two trips to memory overlap)
    {
        const int64_t u0 = (int64_t)gb * 64;
        for (int f = threadIdx.x; f < 64 * F4; f += 64 * SF_NW) {
            const int ur = f / F4, k4 = f % F4;
            const int64_t u = u0 + ur;
            const float4 v = u < B ? reinterpret_cast<const float4*>(Q + u * D)[k4] : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(qs + ur * QLD + 4 * k4) = v;
        }
    }
    __syncthreads();
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    // the B operands (the users' rows rounded to bf16, as MFMA fragments) stay in LDS: [group][s][lane] x 16 bytes, read back per product --
    // in registers they were 32 of 128, and at 128 registers a second workgroup does not fit a CU beside the first (measured: 256 of the
    // 414 workgroups started at once, the others when those were done)
    if (wv < 2) {
        const float* qrow = qs + (wv * 32 + c) * QLD;
#pragma unroll
        for (int s = 0; s < NS16; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + 16 * s + 8 * h), b = *reinterpret_cast<const float4*>(qrow + 16 * s + 8 * h + 4);
            const b2 p0 = {(__bf16)a.x, (__bf16)a.y}, p1 = {(__bf16)a.z, (__bf16)a.w}, p2 = {(__bf16)b.x, (__bf16)b.y}, p3 = {(__bf16)b.z, (__bf16)b.w};
            bq[(wv * NS16 + s) * 64 + lane] = make_uint4(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1),
                                                         __builtin_bit_cast(unsigned, p2), __builtin_bit_cast(unsigned, p3));
        }
    }
    if (wv < 2) {
        // the exact planes and norms of group wv's 32 users (score_split_k's arithmetic, bit for bit): for the main kernel and the certificate
        const int64_t user = ((int64_t)gb * 2 + wv) * 32 + c;
        const bool uok = user < B;
        const float* qrow = qs + (wv * 32 + c) * QLD;
        double ss = 0.0;
#pragma unroll
        for (int s = 0; s < NS16; ++s) {
            const float4 a = *reinterpret_cast<const float4*>(qrow + 16 * s + 8 * h), b = *reinterpret_cast<const float4*>(qrow + 16 * s + 8 * h + 4);
            const float xv[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            unsigned hi[8], mid[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hi[j] = sx_bf16(xv[j]);
                mid[j] = sx_bf16(xv[j] - __uint_as_float(hi[j] << 16));
                ss += (double)xv[j] * (double)xv[j];
            }
            if (uok) {
                unsigned short* dst = Qs + user * (2 * D) + 16 * s + 8 * h;
                *reinterpret_cast<uint4*>(dst) = make_uint4(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16), hi[4] | (hi[5] << 16), hi[6] | (hi[7] << 16));
                *reinterpret_cast<uint4*>(dst + D) = make_uint4(mid[0] | (mid[1] << 16), mid[2] | (mid[3] << 16), mid[4] | (mid[5] << 16), mid[6] | (mid[7] << 16));
            }
        }
        ss += __shfl_xor(ss, 32, 64);
        float nrm = ss > 1e-30 ? sqrtf((float)ss) : (float)sqrt(ss);
        nrm = nrm * 1.000001f;
        if (ss > 0.0 && nrm < 1.2e-38f) nrm = 1.2e-38f;
        if (h == 0 && uok) qnorm[user] = nrm;
    }
    __syncthreads();
    FRONT_STAMP(1);
    float l[2][8];
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) l[g2][j] = -INFINITY;
    auto insert = [&](float (&ll)[8], float v) {
#pragma unroll
        for (int j = 7; j >= 1; --j) ll[j] = __builtin_amdgcn_fmed3f(v, ll[j], ll[j - 1]);
        ll[0] = fmaxf(v, ll[0]);
    };
    // A starting threshold is a filter value, not a result (a user whose bound came out too high fails the certificate and is redone exactly):
    // the sample is scored with the hi planes alone -- one MFMA per 16 k, |error| <= 2^-7 |q| |e|, a few per cent of the gap between the bound's
    // rank (~ 4 (K + 6)) and the K + 6-th best score it must stay under -- and of every two scores a lane gets only the larger one is inserted
    // (the r-th best of pair maxima is <= the r-th best: the safe side).
    for (int t = wv; t < n_tiles; t += SF_NW) {
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const b2 p0 = {(__bf16)raw[j].x, (__bf16)raw[j].y}, p1 = {(__bf16)raw[j].z, (__bf16)raw[j].w};
            *reinterpret_cast<uint2*>(at + (j * RPI + lrow) * ARS + 2 * lk4) = make_uint2(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1));
        }
        __builtin_amdgcn_sched_barrier(0);                       // (phases kept apart: overlapped by the scheduler they cost 16 more registers than a second workgroup per CU allows)
        if (t + SF_NW < n_tiles) fetch(t + SF_NW);               // (in flight under this tile's products and insertions)
        __builtin_amdgcn_sched_barrier(0);
        // (one group after the other: the second accumulator is 16 registers that decide whether two workgroups fit a CU)
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < NS16; ++s) {
                const bf16x8 H = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(at + c * ARS + 8 * s + 4 * h));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(H, __builtin_bit_cast(bf16x8, bq[(g2 * NS16 + s) * 64 + lane]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; r += 2) insert(l[g2], fmaxf(acc[r], acc[r + 1]));
            __builtin_amdgcn_sched_barrier(0);
        }
        FRONT_STAMP(2 + ((t / SF_NW) & 7));
    }
    FRONT_STAMP(10);
    // the waves' lists of a group are joined by wave `group`; a list that lost entries beyond its 8 only makes the bound lower (safe side)
    __syncthreads();                                             // (the query rows in LDS are done with: the lists go over them)
#pragma unroll
    for (int g2 = 0; g2 < 2; ++g2)
#pragma unroll
        for (int j = 0; j < 8; ++j) lx2[((wv * 2 + g2) * 8 + j) * 64 + lane] = l[g2][j];
    __syncthreads();
    if (wv < 2) {
        float m[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
        for (int w = 0; w < SF_NW; ++w)
#pragma unroll
            for (int j = 0; j < 8; ++j) insert(m, lx2[((w * 2 + wv) * 8 + j) * 64 + lane]);
        float mine = m[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) mine = (j == rhalf - 1) ? m[j] : mine;
        const float bound = fminf(mine, __shfl_xor(mine, 32, 64));
        const int64_t user = ((int64_t)gb * 2 + wv) * 32 + c;
        if (h == 0 && user < B) gthr[user] = bound > -INFINITY ? sr_enc(bound) : 0u;
    }
    FRONT_STAMP(11);
}

// joins the chunks of score_bound_k: one wave per 32 users, lane = (user, half) as there
__global__ __launch_bounds__(64) void score_bound_merge_k(const float* __restrict__ partial, int nchunks, int64_t B, int rhalf,
                                                          unsigned* __restrict__ gthr) {
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    const int64_t user = (int64_t)blockIdx.x * 32 + c;
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) l[j] = -INFINITY;
    for (int ch = 0; ch < nchunks; ++ch) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = partial[(((int64_t)blockIdx.x * nchunks + ch) * 8 + j) * 64 + lane];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int k = 7; k >= 1; --k) l[k] = __builtin_amdgcn_fmed3f(v[j], l[k], l[k - 1]);
            l[0] = fmaxf(v[j], l[0]);
        }
    }
    float mine = l[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) mine = (j == rhalf - 1) ? l[j] : mine;
    const float bound = fminf(mine, __shfl_xor(mine, 32, 64));
    if (h == 0 && user < B && bound > -INFINITY) gthr[user] = sr_enc(bound);
}

// score_topk_merge_x: one wave per user.  The split kernel leaves, per (user, segment, lane half), a list of its C best items
// by APPROXIMATE score s' and the bound t below which it dropped everything else (its final threshold).
// (1) Merge the user's lists into the best 64 by (s', lowest id).  Every item of the catalog that is not among them has
//     s' <= T := max(all t, the 64th merged s' if the lists held more than 64 entries)  (-inf: nothing was ever dropped).
// (2) Re-score the 64 candidates with the exact chain acc = fmaf(q[k], e[k], acc), k = 0..D-1 (the oracle's and the fp32 MFMA
//     kernel's arithmetic) and sort them by (s, lowest id).
// (3) Certificate: with |s' - s| <= eps for every item of this user, an outsider has s <= T + eps; if the K-th best exact
//     candidate score is strictly above T + eps, the exact top K of the whole catalog are the best K candidates -- written
//     out, bit-exact.  Otherwise the user is flagged and the exact kernel redoes the user's block (score_kernel_reg<.., false>
//     with blockflag, score_topk_merge with userflag): correctness never depends on eps being small or on the lists being
//     long enough, only on eps being an upper bound.
//   eps = cerr * |q| * max_j |e_j| (+ 1e-36 for flushed denormals), cerr = 1.05 * (3.01 * 2^-18 [dropped mid.mid, split remainders]
//         + (3 D + 4) * 2^-23 [fp32 accumulation of the 3 D exact bf16 x bf16 products inside the MFMAs, one ulp per addition]
//         + D * 2^-24 [the exact chain's own distance from the true dot product]), using sum |q_k e_k| <= |q| |e|.
#define MX_ROWS 16   // candidate rows staged per pass and wave (4 waves x 16 x (D + 4) floats of LDS; 32 rows per pass measured slower)
__device__ float g_sx_info[8];
__device__ unsigned g_sx_stats[2];   // [0] users sent to the exact fallback so far, [1] diagnostics: max |s' - s| / eps (float bits)
// The split path's fallback for catalogs of up to SX_RESCAN_MAX_N items: ONE wave per FLAGGED user (the list score_topk_merge_x wrote)
// scans the whole catalog with the exact fmaf chain -- 64 items at a time, rows staged through the wave's LDS slice by LDS-DMA as in the
// merge's re-scoring, seen items dropped by a cursor into the user's sorted list, each chunk that holds anything better than the current
// K-th merged into the wave's best-64 list by the merge's bitonic networks -- and writes the user's result.  Nobody flagged (the normal case):
// every wave reads one word and leaves; that replaces the TWO launches (exact kernel over flagged blocks + its merge) whose dispatch was
// 9 of the call's 387 us.  A flagged user costs ~N / 64 chunk rounds of ~2 000 cycles on one wave (Beauty: ~0.15 ms; every user of a
// 22 363-user call flagged -- all-zero queries: ~2 ms instead of the exact kernels' 0.6): correctness, as before, never depends on it.
#define SX_RESCAN_MAX_N (1 << 17)
// one flagged user, by one wave: `stage` = the wave's MX_ROWS x D floats of LDS
template <int D>
__device__ __forceinline__ void sx_rescan_user(int64_t user, int64_t N, int K, const int64_t* __restrict__ seen_ptr, const int64_t* __restrict__ seen_idx,
                                               const float* __restrict__ Q, const float* __restrict__ E, float* __restrict__ vals,
                                               int64_t* __restrict__ idx, float* stage, int lane) {
    constexpr int LPR = D / 4;
    constexpr int CPR = LPR, RPI = 64 / CPR, IPP = MX_ROWS / RPI;
    const int PAD = 0x7FFFFFFF;
    const float* qrow = Q + user * D;
    int64_t sp = seen_ptr ? seen_ptr[user] : 0;
    const int64_t se = seen_ptr ? seen_ptr[user + 1] : 0;
    int64_t wbase = -1, w = 0;
    float bv = -INFINITY;
    int bi = PAD;
    for (int64_t c0 = 0; c0 < N; c0 += 64) {
        float sx = -INFINITY;
#pragma unroll 1
        for (int r0 = 0; r0 < 64; r0 += MX_ROWS) {
            if (c0 + r0 >= N) break;
#pragma unroll
            for (int i = 0; i < IPP; ++i) {
                const int r = i * RPI + lane / CPR;
                int64_t id = c0 + r0 + r;
                if (id >= N) id = N - 1;
                const int g = (lane % CPR) ^ (r & 15);
                const float* src = E + id * D + 4 * g;
                __attribute__((address_space(3))) unsigned char* dst =
                    (__attribute__((address_space(3))) unsigned char*)(__attribute__((address_space(3))) float*)stage + i * 1024;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, dst, 16, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane >= r0 && lane < r0 + MX_ROWS) {
                const int rr = lane - r0;
                const float* erow = stage + rr * D;
                float acc = 0.0f;
#pragma unroll
                for (int k = 0; k < LPR; ++k) {
                    const float4 e = *reinterpret_cast<const float4*>(erow + 4 * (k ^ (rr & 15)));
                    acc = fmaf(qrow[4 * k + 0], e.x, acc);
                    acc = fmaf(qrow[4 * k + 1], e.y, acc);
                    acc = fmaf(qrow[4 * k + 2], e.z, acc);
                    acc = fmaf(qrow[4 * k + 3], e.w, acc);
                }
                sx = acc + 0.0f;   // (-0 -> +0, like the list keys of the exact kernel)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice is rewritten by the next pass
        }
        const int64_t item = c0 + lane;
        bool valid = item < N && sx >= -INFINITY;      // (a NaN score never enters a list: v_cmp_ge in the exact kernel)
        // the user's seen items inside [c0, c0 + 64): a cursor into the ascending list, 64 entries at a time
        for (;;) {
            if (sp >= se) break;
            if (wbase != sp) { w = sp + lane < se ? seen_idx[sp + lane] : (int64_t)0x7FFFFFFFFFFFFFFFll; wbase = sp; }
            unsigned long long inr = __ballot(w < c0 + 64);
            const int cnt = __popcll(inr);
            while (inr) {
                const int b = __ffsll((long long)inr) - 1;
                inr &= inr - 1;
                const int64_t sid = ((int64_t)__shfl((int)(w >> 32), b, 64) << 32) | (unsigned)__shfl((int)w, b, 64);
                if (sid == item) valid = false;
            }
            sp += cnt;
            if (cnt < 64) break;
        }
        float v = valid ? sx : -INFINITY;
        int i = valid ? (int)item : PAD;
        const float kv = __shfl(bv, K - 1, 64);
        const int ki = __shfl(bi, K - 1, 64);
        if (__ballot(valid && (ki == PAD || sc_before(v, i, kv, ki))) == 0ull) continue;
        bitonic_sort64(v, i, lane);
        const float rv = __shfl(v, 63 - lane, 64);
        const int ri = __shfl(i, 63 - lane, 64);
        if (sc_before(rv, ri, bv, bi)) { bv = rv; bi = ri; }
#pragma unroll
        for (int j = 32; j > 0; j >>= 1) bitonic_step(bv, bi, j, (lane & j) == 0, lane);
    }
    topk_emit(bv, bi, lane, user, N, K, seen_ptr, seen_idx, vals, idx);
}

template <int D>
__global__ __launch_bounds__(256) void score_rescan_k(const int* __restrict__ nfl, const int* __restrict__ fl_list, int64_t B, int64_t N, int K,
                                                      const int64_t* __restrict__ seen_ptr, const int64_t* __restrict__ seen_idx,
                                                      const float* __restrict__ Q, const float* __restrict__ E,
                                                      float* __restrict__ vals, int64_t* __restrict__ idx) {
    __shared__ __align__(16) float mx_stage[4 * MX_ROWS * D];
    int n = *nfl;
    if (n <= 0) return;
    if (n > B) n = (int)B;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* stage = mx_stage + wv * (MX_ROWS * D);
    for (int f = blockIdx.x * 4 + wv; f < n; f += gridDim.x * 4)
        sx_rescan_user<D>(__builtin_amdgcn_readfirstlane(fl_list[f]), N, K, seen_ptr, seen_idx, Q, E, vals, idx, stage, lane);
}

template <int D>
#ifndef MX_WAVES64
#define MX_WAVES64 8     // waves per SIMD the D = 64 merge is compiled for: 64 registers and 76 B of scratch instead of 86 and none -- 5 -> 8 resident
                         // waves per SIMD; the launch is a chain of dependent steps per user: 0.4038 -> 0.3998 ms per call (6: 0.4066; scripts/score_ab.py)
#endif
__global__ __launch_bounds__(256, D == 64 ? MX_WAVES64 : 4) void score_topk_merge_x(const float* __restrict__ part_vals, const int* __restrict__ part_idx,
                                                          const float* __restrict__ part_T,
                                                          int maxseg, int64_t B, int64_t N, int K, int C, int64_t nst, int64_t upw,
                                                          const int64_t* __restrict__ seen_ptr, const int64_t* __restrict__ seen_idx,
                                                          const float* __restrict__ Q, const float* __restrict__ E,
                                                          const float* __restrict__ qnorm, const unsigned* __restrict__ emax, float cerr,
                                                          float* __restrict__ vals, int64_t* __restrict__ idx,
                                                          int* __restrict__ userflag, int* __restrict__ blockflag,
                                                          int dbg_maxerr, int segs, int n_emax, int* __restrict__ fl_list, int rescan_here) {
    constexpr int LPR = D / 4;
    __shared__ __align__(16) float mx_stage[4 * MX_ROWS * D];
    const int mxd = dbg_maxerr >> 4;
    dbg_maxerr &= 15;
    const int lane = threadIdx.x & 63;
    const int64_t user = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (user >= B) return;
    const int64_t ub = user / SC_USERS;
    const int64_t w0 = (ub * nst) / upw, w1 = ((ub + 1) * nst - 1) / upw;
    const int nseg = segs > 0 ? segs : (int)(w1 - w0 + 1);   // one pair list per segment
    const int PAD = 0x7FFFFFFF;
    float bv = -INFINITY;
    int bi = PAD;
    float T = -INFINITY;
    int total = 0;
    // Four lists per round, the NEXT round's loads issued before this round's merge network: with one list per iteration the loop is a
    // chain of global memory round trips, and that chain, not the network, is this kernel's time (24 lists per user for a
    // 512-user batch against Beauty, 256 for one against a 12.5 M-item shard).
    float v4[4], t4[4], nv4[4], nt4[4];
    int i4[4], ni4[4];
    auto fetch4 = [&](int s0, float* fv, int* fi, float* ft) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u;
            fv[u] = -INFINITY; fi[u] = -1; ft[u] = -INFINITY;
            if (s < nseg) {
                if (lane < C) {   // (round 6: non-temporal loads here -- to leave the L2 to the item rows gathered below -- changed nothing: 59.6 vs 58.8 us)
                    fv[u] = part_vals[(user * maxseg + s) * C + lane];
                    fi[u] = part_idx[(user * maxseg + s) * C + lane];
                }
                ft[u] = part_T[user * maxseg + s];
            }
        }
    };
    const int nround = (mxd & 1) ? 1 : nseg;
    fetch4(0, v4, i4, t4);
    for (int s0 = 0; s0 < nround; s0 += 4) {
        fetch4(s0 + 4 < nround ? s0 + 4 : nseg, nv4, ni4, nt4);   // (past the end: nothing is loaded)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float v = v4[u];
            int i = i4[u];
            if (i < 0) { i = PAD; v = -INFINITY; }
            T = fmaxf(T, t4[u]);
            total += __popcll(__ballot(i != PAD));
            const float rv = __shfl(v, 63 - lane, 64);
            const int ri = __shfl(i, 63 - lane, 64);
            if (sc_before(rv, ri, bv, bi)) { bv = rv; bi = ri; }
#pragma unroll
            for (int j = 32; j > 0; j >>= 1) bitonic_step(bv, bi, j, (lane & j) == 0, lane);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { v4[u] = nv4[u]; i4[u] = ni4[u]; t4[u] = nt4[u]; }
    }
    // Of the (up to) 64 merged entries the best NC = K + 6 rounded up to a multiple of 8 (56 at K = 50) are re-scored; the rest count as
    // dropped, the best of them -- lane NC -- into T.  (The pass is bound by the candidates' row gathers from L2, 64 x 4 D bytes per user;
    // the exact K-th then needs to clear the approximate (NC + 1)-th instead of the 65th: thousands of eps on any scores that are not ties.)
    const int NC = ((K + 6 + 7) & ~7) < 64 ? ((K + 6 + 7) & ~7) : 64;
    if (total > NC && NC < 64) T = fmaxf(T, __shfl(bv, NC, 64));
    else if (total > 64) T = fmaxf(T, __shfl(bv, 63, 64));
    if (lane >= NC) { bi = PAD; bv = -INFINITY; }
    const int nvalid = total < NC ? total : NC;
    // exact re-scoring: lane = candidate; the candidates' rows come in through the wave's LDS slice, MX_ROWS rows per pass (a lane
    // reading its own row from global memory would touch 64 cache lines per load instruction).
    float* stage = mx_stage + (threadIdx.x >> 6) * (MX_ROWS * D);
    // the query row: the user is the wave's, so the row comes in through scalar loads and the chain multiplies by scalar registers
    const float* qrow = Q + (int64_t)__builtin_amdgcn_readfirstlane((int)user) * D;
    // Rows by LDS-DMA (global_load_lds_dwordx4: no staging registers, no LDS stores): an instruction moves 1 KB = 64 / CPR
    // whole rows into the wave's slice, lane l the 16-byte slot l; rows are unpadded and slot (l % CPR) of row r holds the row's
    // chunk (l % CPR) ^ (r & 15), so that MX_ROWS lanes reading chunk k of their own rows hit MX_ROWS different bank groups.
    float sx = -INFINITY;
    constexpr int CPR = LPR, RPI = 64 / CPR, IPP = MX_ROWS / RPI;
#pragma unroll 1
    for (int r0 = 0; r0 < nvalid && !(mxd & 2); r0 += MX_ROWS) {
#pragma unroll
        for (int i = 0; i < IPP; ++i) {
            if (r0 + i * RPI >= nvalid) break;     // (56 candidates = three and a half passes: the last pass's second half is not fetched)
            const int r = i * RPI + lane / CPR;
            const int id = __shfl(bi, r0 + r, 64);
            const int g = (lane % CPR) ^ (r & 15);
            const float* src = E + (int64_t)(id != PAD ? id : 0) * D + 4 * g;
            __attribute__((address_space(3))) unsigned char* dst =
                (__attribute__((address_space(3))) unsigned char*)(__attribute__((address_space(3))) float*)stage + i * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, dst, 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane >= r0 && lane < r0 + MX_ROWS && bi != PAD) {
            const int rr = lane - r0;
            const float* erow = stage + rr * D;
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < LPR; ++k) {
                const float4 e = *reinterpret_cast<const float4*>(erow + 4 * (k ^ (rr & 15)));
                acc = fmaf(qrow[4 * k + 0], e.x, acc);
                acc = fmaf(qrow[4 * k + 1], e.y, acc);
                acc = fmaf(qrow[4 * k + 2], e.z, acc);
                acc = fmaf(qrow[4 * k + 3], e.w, acc);
            }
            sx = acc + 0.0f;   // (-0 -> +0, like the list keys of the exact kernel)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice is rewritten by the next pass
    }
    // (the item table's largest row norm: one word, or one word per splitting workgroup of score_front_k -- n_emax <= 64; non-negative floats
    //  and the NaN above them order as unsigned words)
    unsigned emw = lane < n_emax ? emax[lane] : 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) emw = max(emw, (unsigned)__shfl_xor((int)emw, o, 64));
    const double eps = (double)cerr * (double)qnorm[user] * (double)__uint_as_float(emw) + 1e-36;
    if (dbg_maxerr && bi != PAD) {   // diagnostics: the largest observed |s' - s| / eps (must stay below 1; tests/test_gpu_ops.py)
        const float ratio = (float)(fabs((double)sx - (double)bv) / eps);
        atomicMax(&g_sx_stats[1], __float_as_uint(ratio));
    }
    // The candidates arrive in the order of their approximate scores, which is almost always the exact order too: sort only if
    // some neighbour pair is out of order (a near-tie inside the error bound: a few users in a hundred).
    {
        const float nv = __shfl_down(sx, 1, 64);
        const int ni = __shfl_down(bi, 1, 64);
        const bool inverted = lane < 63 && sc_before(nv, ni, sx, bi);
        if (__ballot(inverted) != 0ull && !(mxd & 4)) bitonic_sort64(sx, bi, lane);
    }
    const float xk = __shfl(sx, K - 1, 64);
    const bool validk = __shfl(bi, K - 1, 64) != PAD;
    const bool pass = T == -INFINITY || (validk && (double)xk > (double)T + eps) || mxd != 0;   // (T = -inf: nothing was ever dropped)
    if (!pass) {
        if (lane == 0) {
            userflag[user] = 1; blockflag[ub] = 1; atomicAdd(&g_sx_stats[0], 1u);
            // "somebody is flagged": the word the fallback kernel looks at first -- and, where the call ends in score_rescan_k (fl_list), the
            // NUMBER of flagged users, each of them listed
            int* const nfl = &blockflag[(B + SC_USERS - 1) / SC_USERS + 1];
            if (fl_list) fl_list[atomicAdd(nfl, 1)] = (int)user;
            else *nfl = 1;
            if (dbg_maxerr) {   // diagnostics: the last flagged user's certificate inputs
                g_sx_info[0] = (float)user; g_sx_info[1] = T; g_sx_info[2] = xk; g_sx_info[3] = (float)eps;
                g_sx_info[4] = (float)total; g_sx_info[5] = (float)nseg; g_sx_info[6] = validk ? 1.f : 0.f; g_sx_info[7] = (float)K;
            }
        }
        // (the fallback right here, by the wave that found it necessary: the separate launch of waves per flagged user did nothing else, and
        //  nobody flagged -- the normal case -- was a launch of ~3 us at the end of every call)
        if (rescan_here) sx_rescan_user<D>(user, N, K, seen_ptr, seen_idx, Q, E, vals, idx, stage, lane);
        return;
    }
    topk_emit(sx, bi, lane, user, N, K, seen_ptr, seen_idx, vals, idx);
}

// ---------------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------------
// Tuning / diagnostic switches.  In the product library (librecengine.so) they are compile-time constants and no re_dbg_* symbol
// exists: the C ABI has no global mutable state (include/recengine.h).  `make dbg` (-DRE_DEBUG) builds librecengine_dbg.so, in
// which the same switches are variables behind re_dbg_* hooks -- for the A/B tests (tests/test_gpu_score_split.py) and the
// tuning scripts.
#ifdef RE_DEBUG
#define RE_SWITCH static
#else
#define RE_SWITCH static const
#endif
RE_SWITCH int g_score_pop = 3;      // tuning switches (scripts/tune_score.py); not part of the ABI
RE_SWITCH int64_t g_score_minseg = SC_MIN_SEG;
RE_SWITCH int64_t g_score_maxwgs = SC_MAX_WGS;
#ifdef RE_DEBUG
extern "C" void re_dbg_score_maxwgs(int64_t n) { g_score_maxwgs = n > 0 ? n : SC_MAX_WGS; }
#endif
RE_SWITCH int g_score_share = 1;   // workgroups share per-user bounds through global memory (0: A/B switch, scripts/tune_score.py)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_share(int on) { g_score_share = on; }
#endif
RE_SWITCH int g_score_dbg = 0;   // diagnostics only (scripts/tune_score.py): 1 = no hits at all, 2 = append but never insert
#ifdef RE_DEBUG
extern "C" void re_dbg_score_diag(int mode) { g_score_dbg = (g_score_dbg & ~0xFF) | (mode & 0xFF); }
#endif
#ifdef RE_DEBUG
extern "C" void re_dbg_score_vote(int at) { g_score_dbg = (g_score_dbg & 0xFF) | ((at >= 0 ? at + 1 : 0) << 8); }
#endif
#if defined(SC_PROFILE)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_counters_x(unsigned long long* out2) { (void)hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_sr_counters_x), 16); }
#endif
#endif
#ifdef RE_DEBUG
extern "C" void re_dbg_score_counters(unsigned long long* out4, int reset) {
    (void)hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_sr_counters), 32);
    if (reset) { unsigned long long z[4] = {0, 0, 0, 0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sr_counters), z, 32); }
}
#endif
#ifdef RE_DEBUG
extern "C" void re_dbg_score_variant(int pop, int64_t minseg) { g_score_pop = pop; g_score_minseg = minseg; }
#endif

struct ScorePlan {
    int64_t nub, nst, units, upw;
    int nwg, maxseg;
    bool small = false;   // few users against a short catalog: register-list kernels on segments of >= 8 stages (score_plan_topk)
    int segs = 0;         // > 0: sliced split (score_kernel_reg), `segs` slices of upw stages per user block
};

static ScorePlan score_plan(int64_t B, int64_t N, int64_t D = 64, int64_t wg_cap = 0, int64_t seg_floor = 0) {
    ScorePlan p;
    p.nub = re_cdiv(B, SC_USERS);
    p.nst = re_cdiv(N, SC_TI);
    p.units = p.nub * p.nst;
    // Equal unit ranges per workgroup, at most 2 workgroups per CU -- but never fewer than SC_MIN_SEG stages per
    // segment: every segment re-warms its users' top-K lists (~K(1+ln(T/K)) heap inserts for T items), so slicing a
    // small catalog over all CUs costs more in warm-ups and list merging than it gains (B=512 x N=12101: 0.58 -> see
    // scripts/tune_score.py).
    // (D = 128: the register-list kernel holds a 64-register query fragment and runs one workgroup per CU)
    int64_t upw = re_cdiv(p.units, wg_cap > 0 ? wg_cap : D == 128 && g_score_maxwgs == SC_MAX_WGS ? SC_MAX_WGS / 2 : g_score_maxwgs);
    const int64_t want_seg = seg_floor > g_score_minseg ? seg_floor : g_score_minseg;
    const int64_t min_seg = p.nst < want_seg ? p.nst : want_seg;
    if (upw < min_seg) upw = min_seg;
    p.upw = upw;
    p.nwg = (int)re_cdiv(p.units, p.upw);
    p.maxseg = (int)(re_cdiv(p.nst, p.upw) + 1);
    return p;
}

static size_t score_lds_bytes(int D, int K, bool topk) {
    size_t b = (size_t)SC_TI * (D + 4) * 4;
    if (topk) b += (size_t)K * SC_USERS * 8 + SC_USERS * 4;
    return b;
}

RE_SWITCH int g_score_nb3 = 1;            // three stage buffers for long slices (0: A/B switch)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_nb3(int on) { g_score_nb3 = on; }
#endif
RE_SWITCH int g_score_sliced = 1;         // sliced split for calls with few user blocks (score_plan_topk; 0: A/B switch)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_sliced(int on) { g_score_sliced = on; }
#endif
RE_SWITCH int g_score_small = 1;          // small batches on the register-list kernels (score_plan_topk; 0: A/B switch)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_small(int on) { g_score_small = on; }
#endif
RE_SWITCH int64_t g_score_reg_nub = 16;   // register-list kernels from this many user blocks on (tuning switch, scripts/x2_small.py)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_reg_nub(int64_t n) { g_score_reg_nub = n; }
#endif
RE_SWITCH int g_score_x2 = 1;      // the split (bf16 hi/mid on the XDL pipe + exact re-scoring) fast path; 0 = exact kernel only
#ifdef RE_DEBUG
extern "C" void re_dbg_score_x2(int on) { g_score_x2 = on; }
#endif
RE_SWITCH int g_score_x2_d128 = 1;  // split form at D = 128 (one workgroup per CU: two 32 KB stage buffers; A/B switch)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_x2_d128(int on) { g_score_x2_d128 = on; }
#endif
RE_SWITCH int g_score_sample = 1;   // split form: starting thresholds from a catalog sample (0: A/B switch, scripts/x2_check.py)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_sample(int on) { g_score_sample = on; }
#endif
RE_SWITCH int g_score_front = 1;    // split form: query split + item split + starting thresholds as ONE launch (score_front_k; 0: the three launches)
#ifdef RE_DEBUG
extern "C" void re_dbg_score_front(int on) { g_score_front = on; }
#endif
RE_SWITCH int g_score_rescan = 2;    // the split path's fallback: 2 inside the merge (sx_rescan_user), 1 as score_rescan_k behind it, 0 the exact kernel over flagged blocks + its merge
#ifdef RE_DEBUG
extern "C" void re_dbg_score_rescan(int on) { g_score_rescan = on; }
#endif
RE_SWITCH int g_score_mxdiag = 0;   // timing-only ablation of score_topk_merge_x (scripts/x2_diag.py): 1 no list merge, 2 no re-scoring, 4 no final sort
#ifdef RE_DEBUG
extern "C" void re_dbg_score_mxdiag(int m) { g_score_mxdiag = m; }
#endif
RE_SWITCH int g_score_maxerr = 0;   // diagnostics: record max |s' - s| / eps over all re-scored candidates
#ifdef RE_DEBUG
extern "C" void re_dbg_score_x2_maxerr(int on) { g_score_maxerr = on; }
#endif
#ifdef RE_DEBUG
extern "C" void re_dbg_score_x2_info(float* out8) { (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_sx_info), 32); }
#endif
#ifdef RE_DEBUG
extern "C" void re_dbg_score_x2_stats(unsigned* out2, int reset) {   // (synchronises; tests and scripts only)
    (void)hipMemcpyFromSymbol(out2, HIP_SYMBOL(g_sx_stats), 8);
    if (reset) { unsigned z[2] = {0u, 0u}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sx_stats), z, 8); }
}
#endif
// List capacity of the split form: the smallest instantiated capacity (16, 32, 56) >= K + 6.  A lane's list must be able to
// hold MORE than the user's whole top K: the best K of a user routinely sit in one list (popular items have neighbouring ids
// -- with the bench's Zipf-by-id popularity the best 50 of every trained user are inside the first 64 ids: one stage, two
// lanes), and a list that overflows with them raises its dropped-below bound above the K-th best score, which fails the
// certificate.  Tried on the way: lists as long as a lane's statistical share of the top K (capacity 24: 0.53 ms
// instead of 0.72 on iid scores, every user in the fallback on the trained state); class-strided stage order to spread id
// runs over the segments (a run inside one 64-item stage still lands in two lanes, and the seen cursor then walks every id
// of the user in every segment); 32-entry lists under a pair-level bound with an overflow log for what falls off (the
// pair's K + 6 best split 28 +- 4 between the lanes: the log floods).
static int score_x2_capacity(const ScorePlan& p, int64_t K) {
    (void)p;
    return K + 6 <= 16 ? 16 : K + 6 <= 32 ? 32 : 56;
}
#define SX_MAX_PREP_BYTES (16ll << 30)   // re_score_topk splits the table into its workspace only up to this size (else: exact path)
static bool score_reg_eligible(int64_t N, int64_t D, int64_t K) {
    return g_score_pop == 3 && (D == 64 || D == 128) && K <= 52 && N < (1ll << SR_TAGBITS) - 1;
}
// The plan of a top-K call.  An evaluation batch of a few hundred users against a short catalog (FreeRec's Coach.evaluate scores
// one data-loader batch per call) has too few (user block, stage) units for stream-K over 512 workgroups to leave segments of any
// length: the register-list kernels then run on segments of >= 8 stages -- fewer workgroups, but 113 us instead of the LDS-heap
// kernel's 859 us at 256 x 12 101, 116 instead of 577 us at 512, 120 instead of 521 us at 1 024 (scripts/x2_small.py).
static ScorePlan score_plan_topk(int64_t B, int64_t N, int64_t D, int64_t K) {
    ScorePlan p = score_plan(B, N, D);
    if (!score_reg_eligible(N, D, K)) return p;
    if (g_score_sliced && p.nub <= 64 && p.nst >= 64) {
        // few user blocks: the sliced split -- every block's catalog in the same slices (>= 8 stages each), slice-mates of different
        // blocks on one XCD (see score_kernel_reg)
        const int64_t wgs = D == 128 ? SC_MAX_WGS / 2 : SC_MAX_WGS;
        int64_t segs = wgs / p.nub;
        if (segs > p.nst / 8) segs = p.nst / 8;
        if (segs >= 8) {
            p.upw = re_cdiv(p.nst, segs);
            p.segs = (int)re_cdiv(p.nst, p.upw);
            p.maxseg = p.segs;
            p.nwg = (int)(8 * p.nub * re_cdiv(p.segs, 8));
            p.small = true;
            return p;
        }
    }
    if (g_score_small && p.nub < g_score_reg_nub && p.upw < 64) {
        p = score_plan(B, N, D, 0, 8);
        p.small = true;
    }
    return p;
}
static bool score_use_reg(const ScorePlan& p, int64_t N, int64_t D, int64_t K) {
    return score_reg_eligible(N, D, K) && (p.nub >= g_score_reg_nub || p.upw >= 64 || p.small);
}
static bool score_use_x2(const ScorePlan& p, int64_t N, int64_t D, int64_t K) {
    return g_score_x2 && (D == 64 || (D == 128 && g_score_x2_d128)) && score_use_reg(p, N, D, K) && K + 6 <= 56;
}
static float score_cerr(int64_t D) {
    const double c = 3.01 * ldexp(1.0, -18) + (3.0 * (double)D + 4.0) * ldexp(1.0, -23) + (double)D * ldexp(1.0, -24);
    return (float)(1.05 * c);
}

struct ScoreWs {   // carving of re_score_topk's workspace
    size_t half, off_pi, off_gthr, off_flags, off_qnorm, off_pt, off_bp, off_qs, off_prep, off_fl, total;
    size_t n_zero;   // bytes from off_gthr that are zeroed per call: gthr[B] gthr2[B] userflag[B] blockflag[nub] emax/dbg[64]
};
static ScoreWs score_ws(int64_t B, int64_t N, int64_t D, int64_t K, const ScorePlan& p, bool x2, bool own_prep) {
    ScoreWs w;
    const int64_t kk = x2 ? 56 : K;   // (x2: list capacity <= 56, and the exact fallback's K <= 50)
    int64_t segs = p.maxseg;
    if (x2) { const ScorePlan pfb = score_plan_topk(B, N, D, K); if (pfb.maxseg > segs) segs = pfb.maxseg; }
    w.half = re_align((size_t)p.nub * SC_USERS * segs * 2 * kk * 4);     // up to 2 lists per (user, segment)
    w.off_pi = w.half;
    w.off_gthr = 2 * w.half;
    w.n_zero = re_align((size_t)B * 4 * (x2 ? 3 : 1) + (x2 ? (size_t)p.nub * 4 + 256 : 0));
    w.off_flags = w.off_gthr + (size_t)B * 8;
    w.off_qnorm = w.off_gthr + w.n_zero;
    w.off_pt = w.off_qnorm + (x2 ? re_align((size_t)B * 4) : 0);                     // one dropped-below bound per list
    w.off_bp = w.off_pt + (x2 ? re_align((size_t)p.nub * SC_USERS * segs * 2 * 4) : 0);          // chunk lists of score_bound_k
    w.off_qs = w.off_bp + (x2 ? re_align((size_t)(re_cdiv(B, 32) + 1024) * 64 * 8 * 4) : 0);
    w.off_prep = w.off_qs + (x2 ? re_align((size_t)B * D * 4) : 0);
    w.off_fl = w.off_prep + (x2 && own_prep ? re_align((size_t)N * D * 4) + 256 : 0);       // the flagged users' list (score_rescan_k)
    w.total = w.off_fl + (x2 ? re_align((size_t)B * 4) : 0) + 256;
    return w;
}

extern "C" size_t re_score_topk_workspace_bytes(int64_t B, int64_t N, int64_t D, int64_t K) {
    if (B <= 0 || N <= 0 || K <= 0) return 256;
    ScorePlan p = score_plan_topk(B, N, D, K);
    const bool x2 = score_use_x2(p, N, D, K) && (int64_t)N * D * 4 <= SX_MAX_PREP_BYTES;
    return score_ws(B, N, D, K, p, x2, true).total;
}
// workspace of the prepared form: the item table's split planes live in the caller's `prep` buffer instead
extern "C" size_t re_score_topk_prepared_workspace_bytes(int64_t B, int64_t N, int64_t D, int64_t K) {
    if (B <= 0 || N <= 0 || K <= 0) return 256;
    ScorePlan p = score_plan_topk(B, N, D, K);
    return score_ws(B, N, D, K, p, score_use_x2(p, N, D, K), false).total;
}
extern "C" size_t re_score_prepare_bytes(int64_t N, int64_t D) {
    if (N <= 0 || D <= 0) return 256;
    return re_align((size_t)N * D * 4) + 256;   // planes + the maximum row norm (last 256 bytes)
}

template <int D, bool TOPK, int POP>
static int score_launch(const float* Q, const float* E, int64_t B, int64_t N, const int64_t* seen_ptr, const int64_t* seen_idx,
                        int K, float* pv, int* pi, const ScorePlan& p, float* dense_out, hipStream_t s) {
    const size_t lds = score_lds_bytes(D, K, TOPK);
    auto kern = score_kernel<D, TOPK, POP>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return RE_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(p.nwg), dim3(256), lds, s, Q, E, B, N, seen_ptr, seen_idx, K, pv, pi, p.maxseg, p.nub, p.nst, p.upw, dense_out);
    return re_launch_status();
}

template <bool TOPK>
static int score_dispatch(int64_t D, const float* Q, const float* E, int64_t B, int64_t N, const int64_t* seen_ptr,
                          const int64_t* seen_idx, int K, float* pv, int* pi, const ScorePlan& p, float* dense_out, hipStream_t s) {
    switch (D) {
        case 32: return score_launch<32, TOPK, 2>(Q, E, B, N, seen_ptr, seen_idx, K, pv, pi, p, dense_out, s);
        case 64:
            if (TOPK && g_score_pop == 0) return score_launch<64, TOPK, 0>(Q, E, B, N, seen_ptr, seen_idx, K, pv, pi, p, dense_out, s);
            if (TOPK && g_score_pop == 1) return score_launch<64, TOPK, 1>(Q, E, B, N, seen_ptr, seen_idx, K, pv, pi, p, dense_out, s);
            return score_launch<64, TOPK, 2>(Q, E, B, N, seen_ptr, seen_idx, K, pv, pi, p, dense_out, s);
        case 128: return score_launch<128, TOPK, 2>(Q, E, B, N, seen_ptr, seen_idx, K, pv, pi, p, dense_out, s);
        default: return RE_EUNSUPPORTED;
    }
}

extern "C" int re_score_dense(const float* Q, const float* E, int64_t B, int64_t N, int64_t D, float* out, re_stream_t stream) {
    re_clear_error();
    if (B == 0 || N == 0) return RE_OK;
    if (!Q || !E || !out || B < 0 || N < 0) return RE_EINVAL;
    if ((reinterpret_cast<uintptr_t>(E) & 15u) != 0 || N >= 0x7FFFFFFFll) return RE_EUNSUPPORTED;
    ScorePlan p = score_plan(B, N);
    return score_dispatch<false>(D, Q, E, B, N, nullptr, nullptr, 0, nullptr, nullptr, p, out, (hipStream_t)stream);
}

template <int D>
static int score_split_launch(const float* X, int64_t R, void* Xs, float* rownorm, unsigned* maxnorm, hipStream_t s,
                              void* zfill = nullptr, size_t zfill_bytes = 0) {
    // (8 rounds per workgroup where the table allows: the device-wide maximum costs one atomic per workgroup)
    hipLaunchKernelGGL(score_split_k<D>, dim3(re_grid(R * (D / 4), maxnorm ? 2048 : 256, 2048)), dim3(256), 0, s, X, R, (unsigned short*)Xs, rownorm, maxnorm,
                       (uint4*)zfill, zfill_bytes >> 4);
    return re_launch_status();
}

// Split the item table once for any number of re_score_topk_prepared calls (Coach.evaluate scores every user batch of a split
// against the same table).  prep: re_score_prepare_bytes(N, D) bytes, 16-byte aligned.  Unsupported D -> RE_EUNSUPPORTED.
extern "C" int re_score_prepare(const float* E, int64_t N, int64_t D, void* prep, size_t prep_bytes, re_stream_t stream) {
    re_clear_error();
    if (!E || !prep || N <= 0) return RE_EINVAL;
    if (D != 64 && D != 128) return RE_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(E) & 15u) != 0 || (reinterpret_cast<uintptr_t>(prep) & 15u) != 0) return RE_EUNSUPPORTED;
    if (prep_bytes < re_score_prepare_bytes(N, D)) return RE_EWORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    unsigned* emax = (unsigned*)((char*)prep + re_align((size_t)N * D * 4));
    if (re_zero_async(emax, 256, s) != hipSuccess) return RE_ELAUNCH;
    return D == 64 ? score_split_launch<64>(E, N, prep, nullptr, emax, s) : score_split_launch<128>(E, N, prep, nullptr, emax, s);
}

// prep == NULL: the table is split into the workspace first (re_score_topk); otherwise prep is re_score_prepare's output for E.
static int score_topk_impl(const float* Q, const float* E, int64_t B, int64_t N, int64_t D, const int64_t* seen_ptr,
                           const int64_t* seen_idx, int64_t K, float* vals, int64_t* idx, void* ws, size_t ws_bytes,
                           const void* prep, hipStream_t s) {
    if (B == 0) return RE_OK;
    if (!Q || !E || !vals || !idx || !ws || B < 0 || N <= 0 || K <= 0 || K > RE_TOPK_MAX) return RE_EINVAL;
    if ((reinterpret_cast<uintptr_t>(E) & 15u) != 0 || N >= 0x7FFFFFFFll) return RE_EUNSUPPORTED;
    if (seen_ptr && !seen_idx) return RE_EINVAL;
    if ((reinterpret_cast<uintptr_t>(ws) & 15u) != 0 || (prep && (reinterpret_cast<uintptr_t>(prep) & 15u) != 0)) return RE_EUNSUPPORTED;
    ScorePlan p = score_plan_topk(B, N, D, K);
    const bool x2 = score_use_x2(p, N, D, K) && (prep || (int64_t)N * D * 4 <= SX_MAX_PREP_BYTES) && (reinterpret_cast<uintptr_t>(Q) & 15u) == 0;
    const ScoreWs w = score_ws(B, N, D, K, p, x2, prep == nullptr);
    if (ws_bytes < w.total) return RE_EWORKSPACE;
    float* pv = (float*)ws;
    int* pi = (int*)((char*)ws + w.off_pi);
    unsigned* gthr = (unsigned*)((char*)ws + w.off_gthr);
    int rc, lps = 1;
    const int* userflag = nullptr;
    // (the fallback pass of the split path has its own plan object: same split today -- a flagged user block is redone by as
    // many workgroups as scored it the first time; when nobody is flagged, the normal case, what it costs is its dispatch)
    const ScorePlan pfb = score_plan_topk(B, N, D, K);
    const ScorePlan* lp = &p;
#define SR_LAUNCH(DV, KRV, KTV, X2V, QP, EP, KV, GT, BF, PT) SR_LAUNCH_NB(DV, KRV, KTV, X2V, 2, QP, EP, KV, GT, BF, PT)
#define SR_LAUNCH_NB(DV, KRV, KTV, X2V, NBV, QP, EP, KV, GT, BF, PT)                                                       \
    do {                                                                                                                             \
        auto kern = score_kernel_reg<DV, KRV, KTV, X2V, NBV>;                                                                        \
        const size_t ldsb = (X2V) ? ((NBV) == 3 ? lds_x3 : lds_x2) : lds;                                                           \
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH; \
        hipLaunchKernelGGL(kern, dim3(lp->nwg), dim3(256), ldsb, s, QP, EP, B, N, seen_ptr, seen_idx, (int)(KV), pv, pi, lp->maxseg, lp->nub, \
                           lp->nst, lp->upw, GT, g_score_dbg, BF, PT, lp->segs);                                                                       \
    } while (0)
    // register-list variant: many users per launch (its two lists per segment double the merge work, which dominates
    // when B is small -- A/B in scripts/tune_score.py)
    if (score_use_reg(p, N, D, K)) {
        lps = 2;
        const size_t lds = (size_t)SC_TI * (D + 4) * 4 + (size_t)4 * SR_QC * 64 * 8;
        const size_t lds_x2 = (size_t)2 * SC_TI * D * 4 + (size_t)4 * 20 * 64 * 8;   // split form: two unpadded stage buffers, 20-entry queues
        const size_t lds_x3 = (size_t)3 * SC_TI * D * 4 + (size_t)4 * (D == 64 ? 15 : 20) * 64 * 8;   // ... three, for long slices
        const bool long_slices = g_score_nb3 && p.segs > 0 && p.upw >= 256;   // rows from HBM: request two stages ahead
        if (!x2 && re_zero_async(gthr, w.n_zero, s) != hipSuccess) return RE_ELAUNCH;   // (split form: zeroed by the query split)
        if (!g_score_share) gthr = nullptr;
        const int* blockflag = nullptr;
        if (x2) {
            // ---- fast path: split planes -> approximate scores on the XDL pipe -> exact re-scoring + certificate
            unsigned* gthr2 = (unsigned*)((char*)ws + w.off_gthr) + B;
            int* uflag = (int*)((char*)ws + w.off_flags);
            int* bflag = uflag + B;
            unsigned* emax_own = (unsigned*)(bflag + p.nub);
            float* qnorm = (float*)((char*)ws + w.off_qnorm);
            float* Qs = (float*)((char*)ws + w.off_qs);
            const float* Es;
            const unsigned* emax;
            // starting thresholds from a sample (score_bound_k / score_front_k): n = N/16 items (512 .. 4096), every stride-th row;
            // every 16th item of a short catalog; every 32nd of a long one when there are few enough users for the sample to be
            // cut into chunks over >= 1024 workgroups (a 12.5 M-item shard scored for 512 users: the 4 096-item sample of before let
            // 12 000 items per user through, ~0.5 hits per half tile and wave, so the "quiet" screening rarely applied)
            const int64_t ugroups = re_cdiv(B, 32);
            int64_t nchunks = ugroups >= 1024 ? 1 : 1024 / ugroups;
            int64_t n = N / 16;
            if (nchunks == 1) n = n < 512 ? 512 : n > 4096 ? 4096 : n;
            else n = N / 32 < 4096 ? (N / 16 < 4096 ? N / 16 : 4096) : N / 32;
            if (n < 512) n = 512;
            if (n > N) n = N;
            n &= ~31ll;
            const int64_t stride = n > 0 ? N / n : 1;
            const double m = (double)(K + 6) * (double)n / (double)N;
            int r = (int)ceil(m + SC_BOUND_SIGMAS * sqrt(m) + 2.0);
            r += r & 1;
            const bool sample = gthr && g_score_sample && n >= 32 && r <= 16;
            int n_tiles = (int)(n / 32);
            if (nchunks > n_tiles / 8) nchunks = n_tiles / 8 > 0 ? n_tiles / 8 : 1;
            const int chunk_tiles = n_tiles > 0 ? (int)re_cdiv(n_tiles, nchunks) : 1;
            if (n_tiles > 0) nchunks = re_cdiv(n_tiles, chunk_tiles);
            int n_emax = 1;
            // ONE front launch (score_front_k) where the sample is not cut into chunks (many users) and the item table -- if it is split here --
            // is short enough for <= 62 splitting workgroups of <= 16 rounds each (their maxima: 62 of the 64 words)
            const int64_t e_rounds = re_cdiv(N * (D / 4), 64 * SF_NW);
            const int nbe = prep ? 0 : (int)(e_rounds < 62 ? e_rounds : 62);
            const int64_t fgroups = re_cdiv(B, 64);                                       // (64 users per workgroup of the front launch)
            if (sample && g_score_front && nchunks == 1 && (prep || e_rounds <= 62 * 16) && fgroups + nbe < 0x7FFFFFFF) {
                float* own = (float*)((char*)ws + w.off_prep);
                // gthr2 [B], userflag [B], blockflag [nub], and the first two of the 64 words behind them: [0] the single-word maximum (unused
                // here), [1] "somebody is flagged" (what the fallback kernel looks at first); the splitting workgroups' maxima live in [2, 2 + nbe)
                unsigned* zero_a = (unsigned*)((char*)ws + w.off_gthr) + B;
                const size_t zero_a_n = (size_t)2 * B + (size_t)p.nub + 2;
                unsigned* const bmax = emax_own + 2;
                unsigned* zero_b = bmax + nbe;                                             // what is left of the 64 words
                const size_t zero_b_n = (size_t)(62 - nbe);
                if (D == 64)
                    hipLaunchKernelGGL(score_front_k<64>, dim3((unsigned)(fgroups + nbe)), dim3(64 * SF_NW), 0, s, Q, E, B, N, (unsigned short*)Qs, qnorm,
                                       (unsigned short*)own, bmax, (int)fgroups, nbe, n_tiles, stride, r / 2, gthr, zero_a, zero_a_n, zero_b, zero_b_n);
                else
                    hipLaunchKernelGGL(score_front_k<128>, dim3((unsigned)(fgroups + nbe)), dim3(64 * SF_NW), 0, s, Q, E, B, N, (unsigned short*)Qs, qnorm,
                                       (unsigned short*)own, bmax, (int)fgroups, nbe, n_tiles, stride, r / 2, gthr, zero_a, zero_a_n, zero_b, zero_b_n);
                if ((rc = re_launch_status()) != RE_OK) return rc;
                if (prep) {
                    Es = (const float*)prep;
                    emax = (const unsigned*)((const char*)prep + re_align((size_t)N * D * 4));
                } else {
                    Es = own;
                    emax = bmax;
                    n_emax = nbe;
                }
            } else {
            // the query split zeroes the call's bound / flag words (n_zero is a multiple of 256 bytes) -- it runs FIRST: the item
            // split's device-wide maximum goes into one of those words
            rc = D == 64 ? score_split_launch<64>(Q, B, Qs, qnorm, nullptr, s, (char*)ws + w.off_gthr, w.n_zero)
                         : score_split_launch<128>(Q, B, Qs, qnorm, nullptr, s, (char*)ws + w.off_gthr, w.n_zero);
            if (rc != RE_OK) return rc;
            if (prep) {
                Es = (const float*)prep;
                emax = (const unsigned*)((const char*)prep + re_align((size_t)N * D * 4));
            } else {
                float* own = (float*)((char*)ws + w.off_prep);
                rc = D == 64 ? score_split_launch<64>(E, N, own, nullptr, emax_own, s) : score_split_launch<128>(E, N, own, nullptr, emax_own, s);
                if (rc != RE_OK) return rc;
                Es = own;
                emax = emax_own;
            }
            if (sample) {
                float* bpart = nchunks > 1 ? (float*)((char*)ws + w.off_bp) : (float*)nullptr;
                const dim3 bgrid((unsigned)ugroups, (unsigned)nchunks);
                if (D == 64) hipLaunchKernelGGL(score_bound_k<64>, bgrid, dim3(128), 0, s, Qs, Es, B, n_tiles, stride, r / 2, gthr, chunk_tiles, bpart);
                else hipLaunchKernelGGL(score_bound_k<128>, bgrid, dim3(128), 0, s, Qs, Es, B, n_tiles, stride, r / 2, gthr, chunk_tiles, bpart);
                if ((rc = re_launch_status()) != RE_OK) return rc;
                if (nchunks > 1) {
                    hipLaunchKernelGGL(score_bound_merge_k, dim3((unsigned)ugroups), dim3(64), 0, s, bpart, (int)nchunks, B, r / 2, gthr);
                    if ((rc = re_launch_status()) != RE_OK) return rc;
                }
            }
            }
            const int C = score_x2_capacity(p, K);
            float* pt = (float*)((char*)ws + w.off_pt);
            const bool rescan = g_score_rescan && N <= SX_RESCAN_MAX_N;                // the fallback as ONE launch of waves per flagged user
            int* fl_list = rescan ? (int*)((char*)ws + w.off_fl) : (int*)nullptr;
#define SX_LAUNCH(DV)                                                                          \
    do {   /* per-lane length = half the pair list's capacity */                               \
        if (C == 16) SR_LAUNCH(DV, 8, 8, true, Qs, Es, 8, gthr, nullptr, pt);                  \
        else if (C == 32) SR_LAUNCH(DV, 16, 16, true, Qs, Es, 16, gthr, nullptr, pt);          \
        else if (long_slices) SR_LAUNCH_NB(DV, 28, 28, true, 3, Qs, Es, 28, gthr, nullptr, pt); \
        else SR_LAUNCH(DV, 28, 28, true, Qs, Es, 28, gthr, nullptr, pt);                       \
    } while (0)
            if (D == 64) SX_LAUNCH(64); else SX_LAUNCH(128);
#undef SX_LAUNCH
            if ((rc = re_launch_status()) != RE_OK) return rc;
            if (D == 64)
                hipLaunchKernelGGL(score_topk_merge_x<64>, dim3((unsigned)re_cdiv(B, 4)), dim3(256), 0, s, pv, pi, pt, p.maxseg, B, N, (int)K, C,
                                   p.nst, p.upw, seen_ptr, seen_idx, Q, E, qnorm, emax, score_cerr(D), vals, idx, uflag, bflag, g_score_maxerr | (g_score_mxdiag << 4), p.segs, n_emax, fl_list, rescan && g_score_rescan == 2 ? 1 : 0);
            else
                hipLaunchKernelGGL(score_topk_merge_x<128>, dim3((unsigned)re_cdiv(B, 4)), dim3(256), 0, s, pv, pi, pt, p.maxseg, B, N, (int)K, C,
                                   p.nst, p.upw, seen_ptr, seen_idx, Q, E, qnorm, emax, score_cerr(D), vals, idx, uflag, bflag, g_score_maxerr | (g_score_mxdiag << 4), p.segs, n_emax, fl_list, rescan && g_score_rescan == 2 ? 1 : 0);
            if ((rc = re_launch_status()) != RE_OK) return rc;
            if (rescan && g_score_rescan == 2) return RE_OK;   // (flagged users were re-scored inside the merge)
            if (rescan) {
                const int* nfl = bflag + p.nub + 1;
                const unsigned rg = (unsigned)(re_cdiv(B, 4) < 256 ? re_cdiv(B, 4) : 256);
                if (D == 64) hipLaunchKernelGGL(score_rescan_k<64>, dim3(rg), dim3(256), 0, s, nfl, fl_list, B, N, (int)K, seen_ptr, seen_idx, Q, E, vals, idx);
                else hipLaunchKernelGGL(score_rescan_k<128>, dim3(rg), dim3(256), 0, s, nfl, fl_list, B, N, (int)K, seen_ptr, seen_idx, Q, E, vals, idx);
                return re_launch_status();
            }
            // ---- fallback pass over flagged user blocks only (normally none: every workgroup returns at once)
            blockflag = bflag;
            userflag = uflag;
            if (gthr) gthr = gthr2;
            lp = &pfb;
        }
        if (D == 64) {
            if (K == 50) SR_LAUNCH(64, 50, 50, false, Q, E, K, gthr, blockflag, nullptr);
            else if (K <= 16) SR_LAUNCH(64, 16, 0, false, Q, E, K, gthr, blockflag, nullptr);
            else if (K <= 32) SR_LAUNCH(64, 32, 0, false, Q, E, K, gthr, blockflag, nullptr);
            else SR_LAUNCH(64, 52, 0, false, Q, E, K, gthr, blockflag, nullptr);
        } else {   // D = 128 (config 5): one workgroup per CU, 512-register budget
            if (K == 50) SR_LAUNCH(128, 50, 50, false, Q, E, K, gthr, blockflag, nullptr); else SR_LAUNCH(128, 52, 0, false, Q, E, K, gthr, blockflag, nullptr);
        }
#undef SR_LAUNCH
#undef SR_LAUNCH_NB
        rc = re_launch_status();
    } else {
        rc = score_dispatch<true>(D, Q, E, B, N, seen_ptr, seen_idx, (int)K, pv, pi, p, nullptr, s);
    }
    if (rc != RE_OK) return rc;
    hipLaunchKernelGGL(score_topk_merge, dim3((unsigned)re_cdiv(B, 4)), dim3(256), 0, s, pv, pi, lp->maxseg, lps, (lps == 2) ? 1 : 0, B, N, (int)K, lp->nst, lp->upw,
                       seen_ptr, seen_idx, vals, idx, userflag, lp->segs);
    return re_launch_status();
}

extern "C" int re_score_topk(const float* Q, const float* E, int64_t B, int64_t N, int64_t D, const int64_t* seen_ptr,
                             const int64_t* seen_idx, int64_t K, float* vals, int64_t* idx, void* ws, size_t ws_bytes,
                             re_stream_t stream) {
    re_clear_error();
    return score_topk_impl(Q, E, B, N, D, seen_ptr, seen_idx, K, vals, idx, ws, ws_bytes, nullptr, (hipStream_t)stream);
}

extern "C" int re_score_topk_prepared(const float* Q, const float* E, const void* prep, int64_t B, int64_t N, int64_t D,
                                      const int64_t* seen_ptr, const int64_t* seen_idx, int64_t K, float* vals, int64_t* idx,
                                      void* ws, size_t ws_bytes, re_stream_t stream) {
    re_clear_error();
    if (!prep) return RE_EINVAL;
    return score_topk_impl(Q, E, B, N, D, seen_ptr, seen_idx, K, vals, idx, ws, ws_bytes, prep, (hipStream_t)stream);
}
