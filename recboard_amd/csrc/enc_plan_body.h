// The batch-preparation launch's two kinds of work as device functions of a PL_NT-thread workgroup (enc_plan.hip: sasrec_batch_prep_k runs
// them for the batch it is given; enc_tail.hip: the tail launch of a training step runs them for the NEXT batch in workgroups that are done
// with the item table -- the preparation depends on the batch alone).
#pragma once
#include <math.h>

#include "enc_common.h"
#include "enc_tile_prep.h"
#include "re_rng.h"

#define PL_NT 1024
#define PL_NW (PL_NT / 64)
#define PL_SIF 16          // sequences a wave of the plan workgroup has in flight while it looks for their first real token
#define PL_LDS_B 8192   // sequences whose span / placement fit the plan workgroup's LDS
#define PL_NCLS 19  // 0..2: long sequences of 4 / 3 / 2 tiles; 3 + (16 - span): short sequences by their EXACT span 16 .. 1

__device__ __forceinline__ int pl_class(int span) {
    if (span > 48) return 0;
    if (span > 32) return 1;
    if (span > 16) return 2;
    return 3 + 16 - span;
}

// Where the short sequences go: COMPLEMENT PAIRING, level by level.  Rows are laid out as units of 16, then 8, then 4, 2, 1 rows.  A unit of
// 16 holds one sequence of 9 .. 16 rows -- and, while there are any, one sequence of exactly the complement 16 - s behind it; what is
// left of the sequences of 1 .. 7 rows goes on to the 8-row units (8, 7 + 1, 6 + 2, 5 + 3, or 5 .. 7 alone), then to the 4-row units
// (4, 3 + 1, 3 alone), 2 and 1.  Power-of-two slots alone (the layout until round 3) waste a fifth of the rows of a Beauty-shaped batch
// (293 tiles for 512 sequences); this lays the same batch into 244 (a perfect packing: 241), in closed form from the per-span counts.
struct PlShort {
    int off16[17], c16[8];   // unit index of the first span-s sequence among the 16-row units (s = 9 .. 16); complements taken at this level (t = 1 .. 7)
    int off8[9], c8[4];      // the same for the 8-row units (s = 5 .. 8; t = 1 .. 3)
    int off4[5], c4;         // 4-row units (s = 3, 4; t = 1)
    int base16, base8, base4, base2, base1, end;   // first compact row of every level
};
__device__ __forceinline__ void pl_short_layout(const int* h /* [17]: sequences per span */, int row0, PlShort& P) {
    int u = 0;
    for (int s = 16; s >= 9; --s) { P.off16[s] = u; u += h[s]; }
    const int U16 = u;
    int h8[9];
    h8[8] = h[8];
    for (int t = 1; t <= 7; ++t) { P.c16[t] = h[16 - t] < h[t] ? h[16 - t] : h[t]; h8[t] = h[t] - P.c16[t]; }
    u = 0;
    for (int s = 8; s >= 5; --s) { P.off8[s] = u; u += h8[s]; }
    const int U8 = u;
    int h4[5];
    h4[4] = h8[4];
    for (int t = 1; t <= 3; ++t) { P.c8[t] = h8[8 - t] < h8[t] ? h8[8 - t] : h8[t]; h4[t] = h8[t] - P.c8[t]; }
    P.off4[4] = 0; P.off4[3] = h4[4];
    const int U4 = h4[4] + h4[3];
    P.c4 = h4[3] < h4[1] ? h4[3] : h4[1];
    P.base16 = row0;
    P.base8 = P.base16 + 16 * U16;
    P.base4 = P.base8 + 8 * U8;
    P.base2 = P.base4 + 4 * U4;
    P.base1 = P.base2 + 2 * h4[2];
    P.end = P.base1 + (h4[1] - P.c4);
}
// first compact row of the short sequence of span s that is number r among the sequences of its span
__device__ __forceinline__ int pl_short_row(const PlShort& P, int s, int r) {
    if (s >= 9) return P.base16 + 16 * (P.off16[s] + r);
    if (s == 8) return P.base8 + 8 * (P.off8[8] + r);
    if (r < P.c16[s]) return P.base16 + 16 * (P.off16[16 - s] + r) + (16 - s);
    r -= P.c16[s];
    if (s >= 5) return P.base8 + 8 * (P.off8[s] + r);
    if (s == 4) return P.base4 + 4 * (P.off4[4] + r);
    if (r < P.c8[s]) return P.base8 + 8 * (P.off8[8 - s] + r) + (8 - s);
    r -= P.c8[s];
    if (s == 3) return P.base4 + 4 * (P.off4[3] + r);
    if (s == 2) return P.base2 + 2 * r;
    return r < P.c4 ? P.base4 + 4 * (P.off4[3] + r) + 3 : P.base1 + (r - P.c4);
}

// Barrier of the plan workgroup.  What crosses waves lives in LDS for batches of up to PL_LDS_B sequences: the barrier then only
// orders LDS traffic and the kernel's global stores stay in flight.  Larger batches keep span / placement in global scratch, which other
// waves read behind the barrier: the full barrier (re_sync_full: every wave's stores drained first -- `__syncthreads()` alone does not).
__device__ __forceinline__ void pl_sync(bool lds_only) {
    if (lds_only) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else re_sync_full();
}

// optional SOURCE of the batch: instead of reading (seq, pos, neg) the launch SAMPLES them -- the SASRec training chain of csrc/sampler.hip
// (re_seq_train_sample: same rows, same draws), so a sampled training step is one preparation launch, not a sampler launch + a
// preparation launch.  ptr == nullptr: the batch is read.
#define PL_STREAM_NEG 0x5EEDu
#define PL_MAX_TRIES 32
struct PlSample {
    const int64_t *ptr, *items, *sorted_items, *order;
    int64_t n_order, b0, N;
    uint32_t seed, step;
    int64_t* users;
};
// row b of the sampled batch: (first item of the window in `items`, number of input positions, the user's CSR start and length)
__device__ __forceinline__ void pl_sample_row(const PlSample& SP, int b, int S, int64_t& base, int& len, int64_t& p0, int64_t& n, int64_t& u) {
    u = -1; base = 0; len = 0; p0 = 0; n = 0;
    if (SP.b0 + b < SP.n_order) {
        u = SP.order[SP.b0 + b];
        p0 = SP.ptr[u];
        n = SP.ptr[u + 1] - p0;
        len = (int)(n - 1 < S - 1 ? n - 1 : S - 1);             // window = the last min(n, S) items (HSTU/sampler.py:28-31), one of them the last target
        if (len < 0) len = 0;
        base = p0 + n - 1 - len;
    }
}

// optional: fold the PREVIOUS step's loss into an epoch accumulator (acc[0] += prev[0] * w) -- the epoch loop's "loss.item() per step" of the
// reference (SASRec/main.py:252-256) as one more word of work in a launch that runs anyway, instead of a launch of its own per step
struct PlLoss {
    const float* prev;
    float* acc;
    float w;
};

struct PlWeights {   // optional extra work of the launch: the one-tile-per-workgroup step's weight fragments (enc_tile_prep.h); nblocks = 0: none
    SasrecParams P;
    int L, nblocks, ns;   // ns = D / 16
    uint32_t* wf;
    unsigned* epoch;
};

// What a step's tail launch prepares the NEXT batch from (enc_tail.hip), left in device memory by the stage launch in front of the step:
// the batch's tensors, or (SP.ptr != nullptr) the sampling source -- then the batch is drawn, not read.  Neither: nothing to prepare.
struct PlMail {
    const int64_t *seq, *pos, *neg;
    PlSample SP;
    unsigned epoch;   // the step's number + 1 (never 0, never the value of two steps ago): what this step's span hand-over flag is set to and waited for
    unsigned pad_;
};
#define PL_MAIL_WORDS 16   // int64 words a mailbox occupies (>= sizeof(PlMail) / 8)
static_assert(sizeof(PlMail) <= PL_MAIL_WORDS * 8, "mailbox size");

// ---- element-wise part of the preparation: copies, valid mask, scatter destination rows.  Job `job` of `njobs` (PL_NT threads each).
__device__ __forceinline__ void pl_elementwise(int job, int njobs, const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                               const int64_t* __restrict__ neg, int B, int S, int64_t* __restrict__ seq_out,
                                               int64_t* __restrict__ pos_out, int64_t* __restrict__ neg_out, uint8_t* __restrict__ valid,
                                               int64_t* __restrict__ rows_all, const PlSample& SP) {
    const int tid = threadIdx.x;
    const int64_t n = (int64_t)B * S;
    for (int64_t i = (int64_t)job * PL_NT + tid; i < n; i += (int64_t)njobs * PL_NT) {
        int64_t s, sp = 0, sq = 0;
        if (SP.ptr) {   // sample position i = (row b, position k) exactly as seq_train_sample_k does
            const int b = (int)(i / S), kpos = (int)(i - (int64_t)b * S);
            int64_t base, p0, nn, u;
            int len;
            pl_sample_row(SP, b, S, base, len, p0, nn, u);
            s = 0;
            const int k = kpos - (S - len);
            if (len > 0 && k >= 0) {
                s = SP.items[base + k] + 1;
                sp = SP.items[base + k + 1];
                const int64_t* sl = SP.sorted_items + p0;
                const uint32_t ctr = (uint32_t)i * PL_MAX_TRIES;
                for (int t = 0; t < PL_MAX_TRIES; ++t) {
                    const uint32_t r = re_rng_u32(SP.seed ^ (SP.step * 0x9E3779B1u), PL_STREAM_NEG, ctr + t);
                    sq = (int64_t)(((uint64_t)r * (uint64_t)SP.N) >> 32);
                    int64_t lo = 0, hi = nn;
                    while (lo < hi) {
                        const int64_t mid = (lo + hi) >> 1;
                        if (sl[mid] < sq) lo = mid + 1; else hi = mid;
                    }
                    if (lo >= nn || sl[lo] != sq) break;
                }
            }
            if (kpos == 0 && SP.users) SP.users[b] = u;
        } else {
            s = seq[i];
        }
        const bool v = s != 0;
        if (seq_out) seq_out[i] = s;
        if (valid) valid[i] = v ? 1 : 0;
        if (pos || SP.ptr) {
            const int64_t p = SP.ptr ? sp : pos[i], q = SP.ptr ? sq : neg[i];
            if (pos_out) { pos_out[i] = p; neg_out[i] = q; }
            if (rows_all) {
                rows_all[i] = s;
                rows_all[n + i] = v ? p + 1 : 0;
                rows_all[2 * n + i] = v ? q + 1 : 0;
            }
        }
    }
}

// ---- the encoder's work plan, by ONE workgroup of PL_NT threads.  L: PL_LDS_BYTES of LDS.
#define PL_OFF_SHORT ((PL_NCLS * PL_NW * 4 + (2 * PL_NCLS + 16 + PL_NW + 5) * 4 + 15) & ~15)
#define PL_OFF_SPAN ((PL_OFF_SHORT + (int)sizeof(PlShort) + 15) & ~15)
#define PL_OFF_PLACE (PL_OFF_SPAN + PL_LDS_B)
#define PL_LDS_BYTES (PL_OFF_PLACE + PL_LDS_B * 4)
// mode 0: the whole plan.  Modes 1 and 2 cut it in two for the step's tail launch (enc_tail.hip), where the plan is the longest job of the
// queue: 1 = the spans alone (phase 1: the only part that reads the batch -- two memory round trips for 512 sequences), left in the plan
// buffer's scratch words with a flag -- the step's epoch (PlMail.epoch), never reset -- behind an agent-scope release; 2 = the rest, by ANOTHER
// workgroup that waits for the flag to show this step's epoch (bounded: when it does not come it computes the spans itself).
#define PL_MODE_ALL 0
#define PL_MODE_SPANS 1
#define PL_MODE_REST 2
__device__ __forceinline__ void pl_plan(const int64_t* __restrict__ seq, int B, int S, int ncu, int max_tiles, int split_long,
                                        int* __restrict__ count, int* __restrict__ plan, const PlSample& SP, unsigned char* L, int mode = PL_MODE_ALL,
                                        unsigned epoch = 1u, int part = 0, int nparts = 1) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#if defined(ENC_PROFILE) && defined(PL_PLAN_KERNEL)
    unsigned long long pl_t[6];
    pl_t[0] = __builtin_amdgcn_s_memtime();
#define PL_STAMP(i) pl_t[i] = __builtin_amdgcn_s_memtime()
#elif !defined(PL_STAMP)
#define PL_STAMP(i) do { } while (0)
#endif
#ifdef TAIL_PROFILE
    PL_STAMP(0);
#endif
    int (*const s_cnt)[PL_NW] = reinterpret_cast<int (*)[PL_NW]>(L);                     // [PL_NCLS][PL_NW]
    int* const s_tot = reinterpret_cast<int*>(L + PL_NCLS * PL_NW * 4);
    int* const s_base = s_tot + PL_NCLS;
    int* const s_lay = s_base + PL_NCLS;
    int* const s_red = s_lay + 16;
    int* const s_cb = s_red + PL_NW;
    int& s_nsplit = s_cb[4];
    PlShort& s_short = *reinterpret_cast<PlShort*>(L + PL_OFF_SHORT);
    unsigned char* const s_span = L + PL_OFF_SPAN;
    int* const s_place = reinterpret_cast<int*>(L + PL_OFF_PLACE);
    const int64_t mt = enc_plan_max_tiles(B, S);
    int* hdr = plan;
    int* items = plan + EP_HDR;
    int2* rowmap = (int2*)(plan + enc_plan_rowmap_word(B, S));
    // span / first compact row of every sequence: in LDS for batches up to PL_LDS_B sequences, else in the plan's scratch words
    const bool in_lds = B <= PL_LDS_B;
    int* g_span = plan + enc_plan_rowmap_word(B, S) + 2 * 16 * mt;
    int* g_place = g_span + B;
    unsigned* g_flag = reinterpret_cast<unsigned*>(g_place + B);      // (first of the buffer's 64 spare words; zero in a fresh buffer)
    const bool to_global = !in_lds || mode == PL_MODE_SPANS;          // where phase 1 leaves the spans
    bool have_spans = false;
    if (mode == PL_MODE_REST) {
        int* const s_got = reinterpret_cast<int*>(L + PL_NCLS * PL_NW * 4) + 2 * PL_NCLS + 16 + PL_NW + 4;     // (= &s_nsplit: written again in phase 2)
        if (tid == 0) {
            int spins = 0;
            // (the flag is an EPOCH, never reset: a spans workgroup that publishes after this one gave up -- it is the grid's last block and need not
            //  be resident while this one waits -- leaves a value no later step waits for; with a 0 / 1 flag reset at the end of this job such a
            //  late publish stayed in the buffer and the buffer's next use read the previous batch's spans)
            while (__hip_atomic_load(g_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch && ++spins < (1 << 16)) __builtin_amdgcn_s_sleep(4);
            *s_got = spins < (1 << 16) ? 1 : 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        have_spans = *s_got != 0;
        __syncthreads();
        if (have_spans && in_lds)
            for (int b = tid; b < B; b += PL_NT) s_span[b] = (unsigned char)g_span[b];
    }
    if (tid < PL_NCLS) s_tot[tid] = 0;
    // 1. span of every sequence: a wave per sequence, lane = position (one coalesced load, a ballot, two scalar bit counts: ~10
    //    instructions per sequence -- an element-wise formulation is VALU-bound on this one CU), 16 sequences in flight per wave;
    //    a sequence without any item is given one explicit pad row
    // (PL_MODE_SPANS of a large batch: `nparts` workgroups take a contiguous share of the sequences each -- a wave has sixteen sequences in flight
    //  and a round is a memory round trip: 4 096 sequences were sixteen rounds, ~140 k cycles, on the one workgroup, and the rest of the plan
    //  waited for them: the longest chain of the tail launch from B = 2 048 on)
    const int b_per = nparts > 1 ? ((B + nparts - 1) / nparts + PL_SIF - 1) / PL_SIF * PL_SIF : B;
    const int b_lo = min(part * b_per, B), b_hi = min(b_lo + b_per, B);
    int nnz = 0;
    if (have_spans) {
        // (phase 1 was another workgroup's: spans in place, hdr[4] and count written)
    } else
    if (SP.ptr) {   // sampled batch: a row's span is its window length (every position of the window is a real item) -- no (seq) to read
        for (int b = b_lo + tid; b < b_hi; b += PL_NT) {
            int64_t base, p0, nn, u;
            int len;
            pl_sample_row(SP, b, S, base, len, p0, nn, u);
            const int span = len > 0 ? len : 1;                       // (a row without items: one explicit pad row, as below)
            if (!to_global) s_span[b] = (unsigned char)span; else g_span[b] = span;
            nnz += len;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nnz += __shfl_xor(nnz, o, 64);
    } else
    for (int b0 = b_lo + wave * PL_SIF; b0 < b_hi; b0 += PL_NW * PL_SIF) {   // (PL_SIF sequences in flight per wave: 32 in flight was measured: no faster)
        int64_t v[PL_SIF];
#pragma unroll
        for (int q = 0; q < PL_SIF; ++q) {   // (clamped, unconditional: a predicated load is waited for on the spot)
            const int b = b0 + q < b_hi ? b0 + q : b_hi - 1;
            v[q] = seq[(int64_t)b * S + (lane < S ? lane : S - 1)];
        }
#pragma unroll
        for (int q = 0; q < PL_SIF; ++q) {
            const unsigned long long m = __ballot(lane < S && v[q] != 0);
            const int first = m ? __builtin_ctzll(m) : S - 1;
            if (lane == 0 && b0 + q < b_hi) {
                if (!to_global) s_span[b0 + q] = (unsigned char)(S - first); else g_span[b0 + q] = S - first;
                nnz += __builtin_popcountll(m);
            }
        }
    }
    PL_STAMP(1);
    if (lane == 0) s_red[wave] = nnz;
    pl_sync(in_lds);
    const bool shared_spans = mode == PL_MODE_SPANS && nparts > 1;
    if (tid == 0 && !have_spans && !shared_spans) {
        int c = 0;
        for (int w = 0; w < PL_NW; ++w) c += s_red[w];
        hdr[4] = c;
        if (count) count[0] = c;
    }
    if (mode == PL_MODE_SPANS) {      // publish: every wave's stores drained, then one release and the flag
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            bool last = true;
            if (shared_spans) {
                // (g_flag[1]: how many parts have arrived; g_flag[2]: their token counts -- integer sums, whatever the order.  The LAST part to
                //  arrive takes both back to zero, writes the totals and publishes; its acquire / release pair orders every part's spans before
                //  the flag)
                int c = 0;
                for (int w = 0; w < PL_NW; ++w) c += s_red[w];
                __hip_atomic_fetch_add(reinterpret_cast<int*>(g_flag) + 2, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                last = __hip_atomic_fetch_add(g_flag + 1, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nparts - 1);
                if (last) {
                    __hip_atomic_store(g_flag + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const int tot = __hip_atomic_exchange(reinterpret_cast<int*>(g_flag) + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    hdr[4] = tot;
                    if (count) count[0] = tot;
                }
            }
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(g_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        return;
    }
    // 2. class totals, then ranks (ballot prefix counts: deterministic), in chunks of PL_NT sequences
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            pl_sync(in_lds);
            if (tid == 0) {
                const int n0 = s_tot[0], n1 = s_tot[1], n2 = s_tot[2];
                const int nlong = n0 + n1 + n2, tlong = 4 * n0 + 3 * n1 + 2 * n2;
                int hs[17];
                hs[0] = 0;
                for (int sp = 1; sp <= 16; ++sp) hs[sp] = s_tot[3 + 16 - sp];
                pl_short_layout(hs, 16 * tlong, s_short);
                const int rs = s_short.end - 16 * tlong;
                const int tshort = (rs + 15) >> 4;
                // Sequences of 3 - 4 tiles SPLIT over two workgroups (kinds 2 / 3, enc_common.h) -- only if then every item of the
                // plan still gets a workgroup of its own (the halves wait for each other: both must be resident)
                int nsplit = (split_long & 1) ? n0 + n1 : 0;
                int G = 1, nshort = 0;
                for (int attempt = 0; attempt < 2; ++attempt) {
                    int avail = ncu - nlong - nsplit;
                    if (avail < 1) avail = 1;
                    G = (tshort + avail - 1) / avail;
                    if (G < 1) G = 1;
                    if (G > max_tiles) G = max_tiles;
                    nshort = (tshort + G - 1) / G;
                    if (nsplit == 0 || nlong + nsplit + nshort <= ncu) break;
                    nsplit = 0;
                }
                s_nsplit = nsplit;
                s_lay[0] = nlong; s_lay[1] = tlong; s_lay[2] = tshort; s_lay[3] = G; s_lay[4] = nshort;
                s_lay[5] = 0; s_lay[6] = 4 * n0; s_lay[7] = 4 * n0 + 3 * n1;                 // first tile of the long classes
                s_lay[8] = 0; s_lay[9] = n0; s_lay[10] = n0 + n1;                            // first item of the long classes
                hdr[0] = nlong + nsplit + nshort; hdr[1] = tlong + tshort; hdr[2] = nlong; hdr[3] = G; hdr[5] = nsplit; hdr[6] = ncu;
                // [7]: 1 = the tile kernels run the step (enc_tile.hip; a long sequence's tiles wait for each other across workgroups).
                // (split_long & 2: the caller forbids it; & 4: the caller insists; split items are the workgroup-per-item kernels' device.)
                // One workgroup per CU either way.  The form of batches up to 2048 possible tiles (enc_tile_looped: a workgroup per tile,
                // the grid covers them): <= 1024 tiles, the long sequences' -- laid out first: the first blocks of the grid -- at most three
                // quarters of the CUs, so that every hand-over partner is resident from the start.  The looped form (resident workgroups,
                // further tiles from a counter: any number of tiles) by speed (scripts/large_batch.py, scripts/long_mix.py): while the long
                // sequences' tiles are at most ~1.5 per workgroup (every further pass of chained tiles costs a chain: 500 long tiles on 256
                // workgroups are 11 % faster on the workgroup-per-item kernel, 310 are 6 % slower) and there are at most ~10 tiles per
                // workgroup in all (Beauty-shaped batches of 1 024 / 2 048 / 8 192: tile kernel +10 % / +24 % / -6 %).
                const int tg = (ncu < 256 ? ncu : 256) * ((split_long & 8) ? 2 : 1);   // resident workgroups (& 8: two per CU)
                // (speed rules, measured: at ONE workgroup per CU long tiles beyond 1.5 x the resident workgroups made the workgroup-per-item kernel the
                //  faster one; at TWO per CU the tile kernels win up to the inbox cap -- B = 6 144: 0.461 vs 0.517 ms, 8 192: 0.589 vs 0.732, round 6.
                //  The looped form cannot deadlock on chains: tickets go out in tile order, so whatever a tile waits for is resident or done.)
                const bool fits = enc_tile_looped(B, S) ? ((split_long & 4) || ((split_long & 8 ? true : 2 * tlong <= 3 * tg) && tlong + tshort <= 10 * tg))
                                                        : (tlong + tshort <= 1024 && ((split_long & 4) || tlong <= tg * 3 / 4));
                hdr[7] = (nsplit == 0 && !(split_long & 2) && fits && tlong + tshort <= ENC_XCH_TILE_CAP) ? 1 : 0;   // (the cap: the inboxes' size, whatever the caller insists on)
                for (int k = 0; k < PL_NCLS; ++k) s_base[k] = 0;
                s_cb[0] = 0; s_cb[1] = n0; s_cb[2] = n0 + n1; s_cb[3] = nlong;       // long class k: s_cb[k] long sequences in front of it
            }
            pl_sync(in_lds);
        }
        for (int b0 = 0; b0 < B; b0 += PL_NT) {
            const int b = b0 + tid;
            const int span = b < B ? (in_lds ? (int)s_span[b] : g_span[b]) : 0;
            const int cls = b < B ? pl_class(span) : -1;
            int rank = 0;
#pragma unroll
            for (int k = 0; k < PL_NCLS; ++k) {
                const unsigned long long m = __ballot(cls == k);
                if (lane == 0) s_cnt[k][wave] = __builtin_popcountll(m);
                if (cls == k) rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            }
            pl_sync(in_lds);
            if (pass == 1 && cls >= 0) {
                for (int w = 0; w < wave; ++w) rank += s_cnt[cls][w];
                rank += s_base[cls];
                if (cls < 3) {
                    const int nt = 4 - cls, t0 = s_lay[5 + cls] + nt * rank;
                    if (s_nsplit && cls < 2) {   // first half: two tiles; second half: the rest, behind all long items
                        items[s_cb[cls] + rank] = t0 | (2 << 24) | (2 << 28);
                        items[s_lay[0] + (cls == 0 ? rank : s_tot[0] + rank)] = (t0 + 2) | ((nt - 2) << 24) | (3 << 28);
                    } else {
                        items[s_cb[cls] + rank] = t0 | (nt << 24) | (1 << 28);
                    }
                }
                // the sequence's first compact row: a long sequence owns the tiles of its item, a short one its place among the units
                const int row = cls < 3 ? 16 * (s_lay[5 + cls] + (4 - cls) * rank) : pl_short_row(s_short, span, rank);
                if (in_lds) s_place[b] = row; else g_place[b] = row;
            }
            pl_sync(in_lds);
            if (tid < PL_NCLS) {
                int c = 0;
                for (int w = 0; w < PL_NW; ++w) c += s_cnt[tid][w];
                if (pass == 0) s_tot[tid] += c; else s_base[tid] += c;
            }
            pl_sync(in_lds);
        }
    }
    PL_STAMP(2);
    // 3. short items, then every compact row exactly once: a sequence's wave writes its whole slot (rows behind the span are
    //    dummies), the rows behind the last slot of the last short tile are dummies too
    const int nlong = s_lay[0], tlong = s_lay[1], tshort = s_lay[2], G = s_lay[3], nshort = s_lay[4];
    for (int i = tid; i < nshort; i += PL_NT) {
        const int t0 = i * G;
        const int nt = (tshort - t0) < G ? (tshort - t0) : G;
        items[nlong + s_nsplit + i] = (tlong + t0) | (nt << 24);
    }
    // every compact row exactly once: all rows of the plan's tiles are dummies first (coalesced), then every sequence writes the rows of
    // its span behind the barrier (a lane per row, eight sequences per wave-instruction: the map is 8 bytes per row, ~4 000 rows)
    const int nrows_all = 16 * (tlong + tshort);
    for (int r = tid; r < nrows_all; r += PL_NT) rowmap[r] = make_int2(-1, 0);
    re_sync_full();    // (with vmcnt(0): the dummies are in place before the real rows overwrite them)
    for (int b0 = (tid >> 3); b0 < B; b0 += PL_NT / 8) {
        const int span = in_lds ? (int)s_span[b0] : g_span[b0];
        const int row = in_lds ? s_place[b0] : g_place[b0];
        for (int off = tid & 7; off < span; off += 8) rowmap[row + off] = make_int2(b0 * S + (S - span) + off, S - span);
    }
#ifdef TAIL_PROFILE
    __syncthreads();
    PL_STAMP(3);
#endif
#if defined(ENC_PROFILE) && defined(PL_PLAN_KERNEL)
    __syncthreads();
    PL_STAMP(3);
    if (tid == 0)
        for (int i = 0; i < 4; ++i) g_plan_marks[i] = pl_t[i] - pl_t[0];
#endif
}
