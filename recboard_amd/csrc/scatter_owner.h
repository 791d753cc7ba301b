// The owner-computes scatter-add of a small dense table (scatter.hip: re_scatter_add_rows_small / re_scatter_adam_rows_small) as a device
// function of a 1024-thread workgroup: scatter_owner_k (scatter.hip) is this and nothing else; enc_tail_k (enc_tail.hip) lets the workgroups
// go on with the encoder's weight-gradient jobs when their rows are done.
// The sort above costs six dependent launches however few rows there are -- 44 us of a 180 us SASRec step whose batch holds
// ~13 000 live contribution rows for a 12 102-row table.  For a table that small the inverted index is not worth building: every
// workgroup OWNS `rpw` destination rows (dealt round-robin), scans ALL keys (int32, L2-resident: 54 KB for the batch above) for the ones
// that fall into its range, and adds their rows -- one launch, no workspace, and the zero fill of the untouched rows comes with it
// (every row of dW is written by its owner).
//   Every table row is owned in TWO column halves by two different workgroups (virtual row 2 r + h): a hot row's contributions are then
//   pulled through two CUs' memory pipes, one 128-byte line each, instead of 256 bytes through one.
//   order of summation (bitwise reproducible): matches are numbered in scan order m = 0, 1, ...; lane group m mod 32 loads match m
//   and adds it into accumulator set (m mod 32) mod 8 of that row (LDS, [8][rpw][D]) -- the four groups of a set one after the
//   other; the eight sets of a row are added in order at the end.  A hot row (Zipf head) is thereby spread over all 32 lane
//   groups of its workgroup.  What remains serial is the owner's memory pipe: ~20 GB/s of scattered 256-byte rows per CU
//   (measured), i.e. ~9 us for the 850 rows the Zipf(1.0) head item collects in a 512-sequence batch; the other 255 workgroups
//   are done in 8 us (scripts/scatter_small_marks.py: clock stamps of a workgroup; build with -DSO_MARKS).
//   keys: `n_regions` runs of `n` keys, run q at keys[q * region_stride ...]; row i of run q is g[(q * region_stride + i) * D ...].
//   n comes from device memory (n_dev[0] * n_mul -- e.g. the batch plan's tile count * 16) so that a captured launch follows the batch.
#pragma once
#include <type_traits>

#include "re_common.h"

#define SO_NT 1024           // threads per workgroup: the scan is vector-instruction bound, 4 waves per SIMD hide each other's latencies
#define SO_CAP 12288         // match list (LDS, 48 KB beside the 96 KB of accumulators): 768 entries a wave; the chunked form flushes it when a chunk might not fit
#define SO_NG 8              // accumulator sets
#define SO_LG (SO_NT / 32)   // lane groups (32 lanes each): 4 per accumulator set
#define SO_INF 16            // row loads a lane group keeps in flight (24 was no faster on the Zipf head's 850 rows and cost 22 spilled registers)
#define SO_KPT 8             // keys per thread and chunk: two 16-byte loads
#ifndef SO_STAMP
#define SO_STAMP(i) do { } while (0)   // (enc_tail.hip's profile build)
#endif
#ifdef SO_MARKS   // diagnostic build: workgroup 0 leaves shader-clock stamps in the padding row of dW
#define SO_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) so_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SO_MARK(i) do { } while (0)
#endif

// HS column splits: a table row is owned in HS pieces of DW = D / HS columns by HS DIFFERENT workgroups (virtual row HS * r + h), so
// a hot row's contributions are pulled through HS memory pipes, DW * 4 bytes each, instead of through one.
struct SoAdam {   // W != nullptr: the owner applies the dense Adam update of its rows instead of (or besides) writing their gradient
    float *W, *m, *v;
    const float* hyper;
    float b1, b2, omb1, omb2, eps, wd;
    const unsigned* gate;   // optional device word: non-zero = leave the parameters and moments alone (enc_tail_k: the step's hand-over error word)
};
__device__ __forceinline__ void so_adam1(const SoAdam& A, float ss, float ib, float g, float& p, float& m, float& v) {   // (adam_vec4_dev's arithmetic)
    re_adam1(p, m, v, g, A.b1, A.b2, A.omb1, A.omb2, ss, ib, A.eps, A.wd);
}

struct SoNoHook {
    __device__ __forceinline__ void issue(int, int) {}
    __device__ __forceinline__ void collect() {}
};

// hook.issue(matches, keys): called once by every thread, workgroup-uniformly, when the LAST batch of contribution-row loads has been issued
// (or there was none); hook.collect(): once, after those rows have been added and BEFORE the owned rows' stores are issued -- enc_tail_k takes
// its ticket of the job queue in between: the counter's round trip runs under the row loads' own, and nothing waits for a store's.
// NG accumulator sets per row (SO_NG = 8: the four lane groups of a set add one after the other; a hot row is spread over eight sets) -- or 2
// for tables whose rows do not fit 256 workgroups at 96 rows each (MF-BPR's 34 464 user + item rows were 1 024 workgroups = four passes over the
// CUs at one 144 KB workgroup per CU: 30 us; with two sets a workgroup holds 384 rows and ONE pass does: scatter.hip: scatter_small_launch).
// A list entry is (local row << RSH) | contribution row: 7 + 25 bits at NG = 8, 9 + 23 at NG = 2.
template <int D, int HS, class Hook = SoNoHook, int NG = SO_NG>
__device__ __forceinline__ void so_body(const float* __restrict__ g, const int32_t* __restrict__ keys, int nreg, int64_t stride,
                                        const int32_t* __restrict__ n_dev, int n_mul, int64_t n_host, int64_t R, int rpw,
                                        int64_t padding_idx, float scale, float* __restrict__ dW, const SoAdam& AD, float* so_acc,
                                        Hook&& hook = Hook{}) {

    constexpr int DW = D / HS, VW = DW / 32;         // columns of a piece; floats per lane: a lane group is 32 lanes
    constexpr int HSH = HS == 1 ? 0 : HS == 2 ? 1 : 2;
    typedef float vt __attribute__((ext_vector_type(VW)));
    // so_acc: [NG][rpw][DW] floats of (dynamic) LDS
    constexpr int RSH = NG == SO_NG ? 25 : 23;
    constexpr uint32_t IMASK = (1u << RSH) - 1u;
    constexpr int PRE = NG == SO_NG ? 1 : 3;         // owned-row elements of a thread whose parameter / moment words are requested at the start
    __shared__ uint32_t s_ent[SO_CAP];               // local row << 25 | contribution index
    __shared__ int s_wsum[SO_NT / 64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, grp = tid >> 5, gl = tid & 31;
    const int64_t n = n_dev ? (int64_t)n_dev[0] * n_mul : n_host;
    // rows are dealt round-robin: workgroup w owns rows w, w + nwg, w + 2 nwg, ... (popular items tend to have neighbouring ids:
    // a contiguous range would hand one workgroup most of the batch)
    // (the grid is a power of two: owner and local row of a key are a mask and a shift)
    // (virtual rows kk = (r << HSH) | h are dealt round-robin; this workgroup's pieces all have h = me & (HS - 1))
    const uint32_t nwg = gridDim.x, me = blockIdx.x, wsh = 31 - __clz((int)nwg), h_me = me & (HS - 1);
    const int64_t VR = R * HS;
    const int rows_here = (int64_t)me < VR ? (int)((VR - 1 - me) / nwg + 1) : 0;
    // The launch is a chain of memory round trips (~2 us each under the launch's own load): what the END of the chain needs and does not depend
    // on the keys is requested here -- the step scalars, the gate word and the thread's own piece of parameter / moment rows (the rows a
    // workgroup owns are a function of its index; nobody else writes them).
    float ad_ss = 0.f, ad_ib = 0.f;
    float4 P0[PRE], M0[PRE], V0[PRE];
#pragma unroll
    for (int i = 0; i < PRE; ++i) P0[i] = M0[i] = V0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AD.W) {
#pragma unroll
        for (int i = 0; i < PRE; ++i) {
            const int e = tid + i * SO_NT;
            const int ec = e < rows_here * (DW / 4) ? e : 0;         // (clamped, unconditional; no rows here: row `me` of the table, in range)
            const int64_t o = ((((int64_t)(ec / (DW / 4)) * nwg + me) >> HSH)) * D + h_me * DW;
            if (rows_here > 0) {                                     // (uniform)
                P0[i] = reinterpret_cast<const float4*>(AD.W + o)[ec % (DW / 4)];
                M0[i] = reinterpret_cast<const float4*>(AD.m + o)[ec % (DW / 4)];
                V0[i] = reinterpret_cast<const float4*>(AD.v + o)[ec % (DW / 4)];
            }
        }
        // (unconditional loads, selected afterwards: a branch on a loaded word is a round trip of its own)
        const unsigned gate_w = *(AD.gate ? AD.gate : reinterpret_cast<const unsigned*>(AD.hyper));
        ad_ss = AD.hyper[0];
        ad_ib = AD.hyper[1];
        ad_ib = (AD.gate && gate_w != 0u) ? 0.f : ad_ib;   // ({0, 0}: the caller gated this step off; gate: a hand-over of this step timed out)
    }
    for (int e = tid; e < NG * rpw * DW / 4; e += SO_NT) reinterpret_cast<float4*>(so_acc)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    int cnt = 0;          // entries in the list (workgroup-uniform)
    bool hooked = false;  // (uniform)
    unsigned m0 = 0;      // matches consumed so far
#ifdef SO_MARKS
    unsigned long long so_t[12] = {};
    int so_i = 2;
#endif
    SO_MARK(0);
    SO_STAMP(0);

    const uint32_t n32 = (uint32_t)n, total = (uint32_t)nreg * n32;   // (< 2^25: checked by the entry point)
    // The regions are scanned as ONE run of nreg * n keys.  Region q's keys and rows start q * stride in: flat index v -> word / row v + q * dlt;
    // a list entry holds that row (25 bits) under the key's local row (7 bits).
    auto region_of = [&](uint32_t v) { return (uint32_t)(v >= n32) + (uint32_t)(v >= 2 * n32) + (uint32_t)(v >= 3 * n32); };   // (nreg <= 4)
    const uint32_t dlt = (uint32_t)stride - n32;
    auto flush = [&](bool last) {
        // Lane group grp takes the list entries j with (m0 + j) % 32 == grp, in increasing j, INF row loads in flight at a time; a hot
        // row (the Zipf head is 10 % of a batch: hundreds of entries for ONE workgroup) is thereby spread over all 32 groups.
        // Neighbouring entries of the same row are added in registers first; then the four groups that share an accumulator set
        // (set = grp % 8) add into it one after the other (barriers in between): a fixed order, so the sums are reproducible.
        const int j0 = (int)((grp - m0) & (SO_LG - 1));
        auto rounds = [&](auto inf_tag) {
            constexpr int INF = decltype(inf_tag)::value;
            for (int jb = 0; jb < cnt; jb += SO_LG * INF) {   // (uniform trip count: the barriers below are workgroup-wide)
                vt v[INF];
                int rw[INF];
#pragma unroll
                for (int u = 0; u < INF; ++u) {
                    const int j = jb + j0 + SO_LG * u;
                    const uint32_t en = s_ent[j < cnt ? j : 0];   // (clamped: a valid entry, its value is not used)
                    rw[u] = j < cnt ? (int)(en >> RSH) : -1;
                    v[u] = reinterpret_cast<const vt*>(g + (int64_t)(en & IMASK) * D + h_me * DW)[gl];
                }
                if (last && jb + SO_LG * INF >= cnt) { hook.issue((int)m0 + cnt, (int)total); hooked = true; }   // (uniform)
#pragma unroll
                for (int u = 0; u + 1 < INF; ++u) {
                    const bool same = rw[u] == rw[u + 1];
                    v[u + 1] += same ? v[u] : vt{};
                    rw[u] = same ? -1 : rw[u];
                }
#pragma unroll
                for (int ph = 0; ph < SO_LG / NG; ++ph) {
                    if ((grp / NG) == ph) {
#pragma unroll
                        for (int u = 0; u < INF; ++u) {
                            if (rw[u] >= 0) {
                                vt* a = reinterpret_cast<vt*>(so_acc + ((int64_t)(grp % NG) * rpw + rw[u]) * DW) + gl;
                                *a += v[u];
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        };
        if (cnt <= SO_LG * 4) rounds(std::integral_constant<int, 4>{});   // (uniform) the usual case: a handful of entries per group
        else rounds(std::integral_constant<int, SO_INF>{});
        m0 += (unsigned)cnt;
        cnt = 0;
    };

    __syncthreads();
    SO_MARK(1);
    // The regions are scanned as ONE run of nreg * n keys, 8192 keys per chunk: a thread takes SO_KPT keys as 16-byte loads (the
    // next chunk's are in flight while this one is ranked: the keys were written by another XCD a moment ago, a chunk is a memory
    // round trip), counts its matches, ONE workgroup scan places them, and the matches go to the list from registers.
    // Match order = (chunk, thread, key): a fixed function of the keys.  (n is a multiple of 4 or the tail is handled by element:
    // a load never straddles two regions.)
    const bool vec = (n32 & 3u) == 0;
    auto load_chunk = [&](uint32_t base, int (&kk)[SO_KPT]) {
#pragma unroll
        for (int u = 0; u < SO_KPT / 4; ++u) {
            const uint32_t v = base + (uint32_t)(u * SO_NT + tid) * 4;
            int4 k4 = make_int4(-1, -1, -1, -1);
            if (vec) {
                if (v < total) { const uint32_t q = region_of(v); k4 = *reinterpret_cast<const int4*>(keys + (int64_t)q * stride + (v - q * n32)); }
            } else {
                int* ke = reinterpret_cast<int*>(&k4);
                for (int e = 0; e < 4; ++e)
                    if (v + e < total) { const uint32_t q = region_of(v + e); ke[e] = keys[(int64_t)q * stride + (v + e - q * n32)]; }
            }
            kk[4 * u] = k4.x; kk[4 * u + 1] = k4.y; kk[4 * u + 2] = k4.z; kk[4 * u + 3] = k4.w;
        }
    };
    const uint32_t R32 = (uint32_t)R, pad32 = (padding_idx >= 0 && padding_idx < R) ? (uint32_t)padding_idx : 0xFFFFFFFFu;
    auto entry_of = [&](int key, uint32_t v) { return (((uint32_t)key >> (wsh - HSH)) << RSH) | (v + region_of(v) * dlt); };
    // ---- the usual case: every WAVE ranks its own sixteenth of the keys into its own 512-entry stretch of the list -- ballots and lane counts,
    // no workgroup barrier inside the scan (the chunked form below, two barriers and a workgroup-wide prefix sum per 8 192 keys, took ~5 k cycles
    // a chunk whatever the keys were: 13 k of the tail launch's ~50 k at B = 512, 70 k of ~210 k at B = 4 096; scripts/tail_phases.py) -- and the
    // sixteen stretches are closed up afterwards.  Match order = (wave, chunk of 512 keys, key slot, lane): a fixed function of the keys.  A wave
    // whose stretch would overflow (the owner of a Zipf head row at a large batch) sends the whole workgroup through the chunked form instead.
    constexpr int SO_WCAP = SO_CAP / (SO_NT / 64);
    static_assert(SO_WCAP % 64 == 0 && SO_WCAP / 64 <= 12, "a wave closes up its stretch from twelve registers a lane");
    const uint32_t per_wave = (total + (SO_NT / 64) * 512u - 1u) / ((SO_NT / 64) * 512u) * 512u;
    bool slow = !vec;        // (uniform; n is a multiple of 16 wherever a plan's tile count sets it)
    if (!slow) {
        const uint32_t w_lo = (uint32_t)wid * per_wave, w_hi = min(w_lo + per_wave, total);
        // (no branch around a load and none between the loads of a pass: behind a divergent branch hipcc waits for EVERY outstanding load --
        //  s_waitcnt vmcnt(0) -- and the next chunk's request stopped overlapping anything; addresses are clamped into the wave's range instead,
        //  and keys past its end never count as matches)
        auto load_w = [&](uint32_t base, int (&kk)[SO_KPT]) {
#pragma unroll
            for (int u = 0; u < SO_KPT / 4; ++u) {
                const uint32_t v = min(base + (uint32_t)(u * 64 + lane) * 4, w_hi - 4u);
                const int4 k4 = *reinterpret_cast<const int4*>(keys + (v + region_of(v) * dlt));
                kk[4 * u] = k4.x; kk[4 * u + 1] = k4.y; kk[4 * u + 2] = k4.z; kk[4 * u + 3] = k4.w;
            }
        };
        int wcnt = 0;        // (wave-uniform)
        bool ovf = false;    // (wave-uniform)
        if (total > 0) SO_STAMP(1);
        // (the scan is vector-instruction bound -- sixteen waves of a CU look at every key of the batch.  Per key: an AND, a compare and a scalar
        //  branch; with 128 owners per row half and 64 lanes, 40 % of the keys have SOME lane that belongs here, so the taken side is kept to a
        //  dozen instructions as well: two range compares and the ballot; the lane's slot and the entry word only where a lane really has a match)
        const uint32_t omask = (nwg >> HSH) - 1u, ome = me >> HSH;
        auto rank = [&](const int (&kk)[SO_KPT], uint32_t base) {
#pragma unroll
            for (int u = 0; u < SO_KPT; ++u) {
                const uint32_t k = (uint32_t)kk[u];
                const bool mine_maybe = (k & omask) == ome;
                if (__ballot(mine_maybe) != 0ull) {           // (uniform)
                    const uint32_t v = base + (uint32_t)((u >> 2) * 64 + lane) * 4 + (u & 3);
                    // (a negative key is >= R as unsigned;  v >= w_hi: a clamped load's repeat of the range's last keys)
                    const bool hit = mine_maybe && k < R32 && k != pad32 && v < w_hi;
                    const unsigned long long hm = __ballot(hit);
                    const int pc = __popcll(hm);
                    if (wcnt + pc > SO_WCAP) ovf = true;
                    else {
                        if (hit) s_ent[wid * SO_WCAP + wcnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u))] = entry_of(kk[u], v);
                        wcnt = __builtin_amdgcn_readfirstlane(wcnt + pc);
                    }
                }
            }
        };
        if (w_lo < w_hi) {   // (uniform)
            // (two register sets in turn: a copy at the end of a pass would wait for the chunk just requested)
            int ka[SO_KPT], kb[SO_KPT];
            load_w(w_lo, ka);
            for (uint32_t base = w_lo; base < w_hi; base += 1024u) {
                load_w(base + 512u, kb);
                rank(ka, base);
                if (base == w_lo) SO_STAMP(5);
                load_w(base + 1024u, ka);
                rank(kb, base + 512u);
            }
        }
        if (lane == 0) s_wsum[wid] = ovf ? -1 : wcnt;
        __syncthreads();
        int tot = 0, below = 0;
#pragma unroll
        for (int w = 0; w < SO_NT / 64; ++w) {
            const int x = s_wsum[w];
            slow |= x < 0;
            tot += x;
            below += w < wid ? x : 0;
        }
        SO_STAMP(6);
        if (!slow) {                                          // (uniform) close the stretches up, in place: read, barrier, write
            uint32_t mine[SO_WCAP / 64];
#pragma unroll
            for (int i = 0; i < SO_WCAP / 64; ++i) mine[i] = 64 * i + lane < wcnt ? s_ent[wid * SO_WCAP + 64 * i + lane] : 0u;
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SO_WCAP / 64; ++i)
                if (64 * i + lane < wcnt) s_ent[below + 64 * i + lane] = mine[i];
            cnt = tot;
        }
    }
    if (slow) {                                               // (uniform) the chunked form: flushes the list whenever it might overflow
        __syncthreads();                                      // (everybody has read s_wsum)
        int kv[SO_KPT], kn[SO_KPT];
        if (total > 0) SO_STAMP(1);
        if (total > 0) load_chunk(0, kv);
        for (uint32_t base = 0; base < total; base += SO_NT * SO_KPT) {
            if (base + SO_NT * SO_KPT < total) load_chunk(base + SO_NT * SO_KPT, kn);   // (in flight while this chunk is ranked)
            unsigned mask = 0;
#pragma unroll
            for (int u = 0; u < SO_KPT; ++u) {
                const uint32_t k = (uint32_t)kv[u];   // (a negative key is >= R as unsigned)
                const bool hit = (k & ((nwg >> HSH) - 1)) == (me >> HSH) && k < R32 && k != pad32;
                mask |= (hit ? 1u : 0u) << u;
            }
            const int c = __popc(mask);
            if (base == 0) SO_STAMP(5);
#ifdef SO_MARKS
            if (so_i < 8) { SO_MARK(so_i); ++so_i; }
#endif
            // ---- exclusive scan of the per-thread counts over the workgroup
            int inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            __syncthreads();                        // (previous chunk's s_wsum readers are done)
            if (lane == 63) s_wsum[wid] = inc;
            __syncthreads();
            int tot = 0, below = 0;
#pragma unroll
            for (int w = 0; w < SO_NT / 64; ++w) {
                const int x = s_wsum[w];
                tot += x;
                below += w < wid ? x : 0;
            }
            if (tot != 0) {                         // (uniform)
                if (cnt + tot > SO_CAP) {           // (uniform) make room
                    flush(false);
                    __syncthreads();
                }
                int off = cnt + below + inc - c;
                if (c) {
#pragma unroll
                    for (int u = 0; u < SO_KPT; ++u) {
                        const bool hit = (mask >> u) & 1u;
                        if (hit) {
                            const uint32_t v = base + (uint32_t)((u >> 2) * SO_NT + tid) * 4 + (u & 3);
                            s_ent[off] = entry_of(kv[u], v);
                        }
                        off += hit ? 1 : 0;
                    }
                }
                cnt += tot;
            }
            if (base == 0) SO_STAMP(6);
#pragma unroll
            for (int u = 0; u < SO_KPT; ++u) kv[u] = kn[u];
        }
    }
    __syncthreads();
    SO_MARK(8);
    SO_STAMP(2);
    flush(true);
    if (!hooked) hook.issue((int)m0, (int)total);
    __syncthreads();
    SO_MARK(9);
    SO_STAMP(3);
    hook.collect();
    // ---- the eight accumulators of every owned row, added in set order; untouched rows come out zero
    for (int e = tid; e < rows_here * (DW / 4); e += SO_NT) {
        const int r = e / (DW / 4), c4 = e % (DW / 4);
        float4 s = reinterpret_cast<const float4*>(so_acc + (int64_t)r * DW)[c4];
#pragma unroll
        for (int gq = 1; gq < NG; ++gq) {
            const float4 t = reinterpret_cast<const float4*>(so_acc + ((int64_t)gq * rpw + r) * DW)[c4];
            s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
        }
        const int64_t kk = (int64_t)r * nwg + me;   // virtual row -> table row kk >> HSH, piece h_me
        const float4 gr = make_float4(s.x * scale, s.y * scale, s.z * scale, s.w * scale);
        const int64_t o = (kk >> HSH) * D + h_me * DW;
        if (dW) reinterpret_cast<float4*>(dW + o)[c4] = gr;
        if (AD.W) {
            if (ad_ib != 0.f) {
                const int pi = (e - tid) / SO_NT;               // (uniform)
                float4 P, M, V;
                if (pi < PRE) {
                    P = P0[0]; M = M0[0]; V = V0[0];
#pragma unroll
                    for (int i = 1; i < PRE; ++i)
                        if (pi == i) { P = P0[i]; M = M0[i]; V = V0[i]; }
                } else { P = reinterpret_cast<float4*>(AD.W + o)[c4]; M = reinterpret_cast<float4*>(AD.m + o)[c4]; V = reinterpret_cast<float4*>(AD.v + o)[c4]; }
                const float ss = ad_ss, ib = ad_ib;
                so_adam1(AD, ss, ib, gr.x, P.x, M.x, V.x); so_adam1(AD, ss, ib, gr.y, P.y, M.y, V.y);
                so_adam1(AD, ss, ib, gr.z, P.z, M.z, V.z); so_adam1(AD, ss, ib, gr.w, P.w, M.w, V.w);
                reinterpret_cast<float4*>(AD.W + o)[c4] = P; reinterpret_cast<float4*>(AD.m + o)[c4] = M; reinterpret_cast<float4*>(AD.v + o)[c4] = V;
            }
        }
    }
    SO_STAMP(4);
#ifdef SO_MARKS
    __syncthreads();
    SO_MARK(10);
    if (blockIdx.x == 0 && threadIdx.x == 0)
        for (int i = 0; i < 11; ++i) dW[i] = (float)(so_t[i] ? (long long)(so_t[i] - so_t[0]) : -1ll);
#endif
}

