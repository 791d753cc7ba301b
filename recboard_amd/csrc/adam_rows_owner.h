// (adam_rows.hip: re_sparse_adam_rows_small; enc_tail.hip lets the workgroups go on with the encoder's weight-gradient jobs afterwards)
// Row-sparse Adam for a SMALL key list against a table of any size, in ONE launch and without a sort: what a training step of the
// large-table engines needs (config 5: ~15 k contribution rows per step against a 100 M x 128 table; the general path,
// re_sparse_adam_rows, is a 3-pass radix sort + segmented sum + fix-up = 9 launches, ~70 us at this size).
//
// Owner computes: SA_NWG workgroups; a key belongs to the workgroup its hash names.  Every workgroup scans the whole key list (60 KB of
// int32 keys at config 5: it comes from L2) twice: first it counts the contributions of every distinct key it owns (an LDS hash table),
// then it keeps the entries of the keys with few contributions as (key, position) words in LDS, orders them by all-pairs rank counting
// (a few dozen entries: no barrier-heavy sorting network) and walks the runs of equal keys -- one wave per distinct row sums the row's
// contributions in position order and applies ONE Adam update to (W, m, v): re_sparse_adam_rows' rule (torch.optim.SparseAdam on the
// touched rows + coupled weight decay, the global step count).  A key with many contributions (a Zipf-head item: 8 % of a config-5
// batch is item 1) never enters the list: all the workgroup's waves sum it straight from the key list, each over a slice of the
// positions (matches compacted by ballots, eight row loads in flight), the slices combined in order.  Results do not depend on the order
// in which entries were collected or rows were scheduled.  A workgroup whose share does not fit its LDS structures (adversarial
// inputs) takes a slow exact path: its distinct keys in ascending order, each summed from the key list the same way.
#pragma once
#include "re_common.h"

#define SA_NWG 256
#define SA_NT 1024
#define SA_NW (SA_NT / 64)
#define SA_CAP 2048
#define SA_RUNS 8         // runs of equal keys a wave works on at a time (their rows are requested together)
#define SA_LONG 48        // keys with more contributions than this are summed by the whole workgroup (<= 64: a listed run fits one ballot)

struct SaParams {
    const float* g;
    const void* keys;
    int n_regions;
    int64_t region_stride;
    const int32_t* n_dev;
    int64_t n_mul, n_host;
    int64_t R, padding_idx;
    float *W, *m, *v;
    float b1, b2, omb1, omb2, step_size, inv_sqrt_bc2, eps, wd;
    const float* hyper;
    int64_t stride;   // floats per table / contribution row (D)
    int coff;         // first column of the piece this workgroup owns (set in the kernel: 0, or 64 for the upper half at D = 128)
};

template <int HS>
__device__ __forceinline__ uint32_t sa_owner(uint32_t k) { return (k * 0x9E3779B1u) >> (HS == 2 ? 25 : 24); }   // 256 owners, or 128 pairs

template <int VPT>
struct SaRow {
    float x[VPT];
};

template <int VPT>
__device__ __forceinline__ SaRow<VPT> sa_load(const float* __restrict__ base, int64_t row, int lane, int64_t stride, int coff) {
    SaRow<VPT> r;
    const float* p = base + row * stride + coff + lane * VPT;
    if (VPT == 2) {
        const float2 t = *reinterpret_cast<const float2*>(p);
        r.x[0] = t.x; r.x[VPT - 1] = t.y;
    } else {
        r.x[0] = p[0];
    }
    return r;
}

template <int VPT>
__device__ __forceinline__ void sa_store(float* __restrict__ base, int64_t row, int lane, const SaRow<VPT>& r, int64_t stride, int coff) {
    float* p = base + row * stride + coff + lane * VPT;
    if (VPT == 2) *reinterpret_cast<float2*>(p) = make_float2(r.x[0], r.x[VPT - 1]);
    else p[0] = r.x[0];
}

// One element of the update with the operation sequence PINNED (explicit fused / rounded operations): the kernel has several copies of
// this arithmetic (unrolled row batches, the whole-workgroup path) and which copy a row meets depends on scheduling -- left to the
// compiler's contraction and packed-math choices the copies differed in the last bit.
__device__ __forceinline__ void sa_adam_one(const SaParams& P, float ss, float ib, float g, float& w, float& mm, float& vv) {
    const float gg = __fmaf_rn(P.wd, w, g);
    mm = __fmaf_rn(P.b1, mm, __fmul_rn(P.omb1, gg));
    vv = __fmaf_rn(P.b2, vv, __fmul_rn(__fmul_rn(P.omb2, gg), gg));
    w = __fsub_rn(w, __fmul_rn(ss, __fdiv_rn(mm, __fmaf_rn(__fsqrt_rn(vv), ib, P.eps))));
}

template <int VPT>
__device__ __forceinline__ void sa_adam(const SaParams& P, int64_t row, int lane, const SaRow<VPT>& G) {
    SaRow<VPT> w = sa_load<VPT>(P.W, row, lane, P.stride, P.coff), mm = sa_load<VPT>(P.m, row, lane, P.stride, P.coff), vv = sa_load<VPT>(P.v, row, lane, P.stride, P.coff);
    const float ss = P.hyper ? P.hyper[0] : P.step_size, ib = P.hyper ? P.hyper[1] : P.inv_sqrt_bc2;
#pragma unroll
    for (int c = 0; c < VPT; ++c) sa_adam_one(P, ss, ib, G.x[c], w.x[c], mm.x[c], vv.x[c]);
    sa_store<VPT>(P.W, row, lane, w, P.stride, P.coff);
    sa_store<VPT>(P.m, row, lane, mm, P.stride, P.coff);
    sa_store<VPT>(P.v, row, lane, vv, P.stride, P.coff);
}

// sum of the contribution rows at positions list[e] (low words), e in [e0, e1), in that order; four loads in flight
template <int VPT>
__device__ __forceinline__ SaRow<VPT> sa_sum(const float* __restrict__ g, const unsigned long long* list, int e0, int e1, int lane, int64_t stride, int coff,
                                             SaRow<VPT> acc) {
    int e = e0;
    for (; e + 4 <= e1; e += 4) {
        const SaRow<VPT> a = sa_load<VPT>(g, (int64_t)(uint32_t)list[e], lane, stride, coff), b = sa_load<VPT>(g, (int64_t)(uint32_t)list[e + 1], lane, stride, coff),
                         c2 = sa_load<VPT>(g, (int64_t)(uint32_t)list[e + 2], lane, stride, coff), d = sa_load<VPT>(g, (int64_t)(uint32_t)list[e + 3], lane, stride, coff);
#pragma unroll
        for (int c = 0; c < VPT; ++c) acc.x[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc.x[c], a.x[c]), b.x[c]), c2.x[c]), d.x[c]);
    }
    for (; e < e1; ++e) {
        const SaRow<VPT> a = sa_load<VPT>(g, (int64_t)(uint32_t)list[e], lane, stride, coff);
#pragma unroll
        for (int c = 0; c < VPT; ++c) acc.x[c] += a.x[c];
    }
    return acc;
}

template <class KeyT>
__device__ __forceinline__ int64_t sa_key(const SaParams& P, int64_t pos) { return (int64_t)reinterpret_cast<const KeyT*>(P.keys)[pos]; }

#define SA_HT 1024        // slots of the per-workgroup table of owned distinct keys
#define SA_KPT 16         // keys a thread loads at a time
#define SA_EMPTY 0xFFFFFFFFu
#define SA_WQ ((2 * SA_CAP * 2) / SA_NW)   // u32 words of the two list arrays per wave while they serve as position queues (512)

__device__ __forceinline__ uint32_t sa_slot0(uint32_t k) { return ((k * 0x85EBCA6Bu) >> 12) & (SA_HT - 1); }

// Sum of the contribution rows of ONE key over the whole key list, by all waves of the workgroup: wave w takes the w-th slice of every
// region's positions, first compacts the positions that hold the key into its queue (coalesced key reads, ballots), then sums those rows
// with eight loads in flight; the waves' partial sums are combined in wave order and the row gets its Adam update.
// Must be called by every thread of the workgroup (barriers inside).
template <int VPT, class KeyT>
__device__ __forceinline__ void sa_heavy(const SaParams& P, uint32_t cur, int64_t rows, uint32_t* queue, float (*part)[64 * VPT], int lane, int wave) {
    SaRow<VPT> acc;
#pragma unroll
    for (int c = 0; c < VPT; ++c) acc.x[c] = 0.f;
    uint32_t* q = queue + wave * SA_WQ;
    const int64_t slice = (rows + SA_NW - 1) / SA_NW;
    const int64_t i0 = (int64_t)wave * slice, i1 = (i0 + slice < rows) ? i0 + slice : rows;
    for (int r = 0; r < P.n_regions; ++r) {
        const int64_t base = (int64_t)r * P.region_stride;
        int64_t i = i0;
        while (i < i1) {
            uint32_t nq = 0;                                 // (wave-uniform)
            for (; i < i1 && nq + 256 <= SA_WQ; i += 256) {        // four windows of keys in flight
                int64_t kk[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) kk[u] = (i + 64 * u + lane < i1) ? sa_key<KeyT>(P, base + i + 64 * u + lane) : -1;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const bool hit = kk[u] == (int64_t)cur;
                    const unsigned long long mask = __ballot(hit);
                    if (hit) q[nq + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = (uint32_t)(base + i + 64 * u + lane);
                    nq += (uint32_t)__builtin_popcountll(mask);
                }
            }
            uint32_t e = 0;
            for (; e + 8 <= nq; e += 8) {
                SaRow<VPT> t[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) t[u] = sa_load<VPT>(P.g, (int64_t)q[e + u], lane, P.stride, P.coff);
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int c = 0; c < VPT; ++c) acc.x[c] += t[u].x[c];
            }
            for (; e < nq; ++e) {
                const SaRow<VPT> t = sa_load<VPT>(P.g, (int64_t)q[e], lane, P.stride, P.coff);
#pragma unroll
                for (int c = 0; c < VPT; ++c) acc.x[c] += t.x[c];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < VPT; ++c) part[wave][lane * VPT + c] = acc.x[c];
    __syncthreads();
    if (wave == 0) {
        SaRow<VPT> t;
#pragma unroll
        for (int c = 0; c < VPT; ++c) t.x[c] = 0.f;
        for (int w = 0; w < SA_NW; ++w)
#pragma unroll
            for (int c = 0; c < VPT; ++c) t.x[c] += part[w][lane * VPT + c];
        sa_adam<VPT>(P, (int64_t)cur, lane, t);
    }
    __syncthreads();
}

// HS = 2 (D = 128): a row is owned in two pieces of 64 columns by two DIFFERENT workgroups (the update is column-wise independent), so a
// Zipf-head row's contributions are pulled through two compute units' memory pipes; every key is then listed by two workgroups.
#define SA_LDS_BYTES(VPT) (2 * SA_CAP * 8 + SA_CAP * 4 + 2 * SA_HT * 4 + SA_NW * 64 * (VPT) * 4)   // the body's arrays (53 248 at VPT = 1)
template <int VPT, int HS, class KeyT>
__device__ __forceinline__ void sa_body(SaParams P, unsigned char* lds) {
    unsigned long long* const s_lists = reinterpret_cast<unsigned long long*>(lds);   // collected entries | ordered entries; the heavy phase's position queues
    unsigned long long* const s_list = s_lists;
    unsigned long long* const s_sorted = s_lists + SA_CAP;
    uint32_t* const s_seg = reinterpret_cast<uint32_t*>(s_lists + 2 * SA_CAP);
    uint32_t* const s_hkey = s_seg + SA_CAP;
    uint32_t* const s_hcnt = s_hkey + SA_HT;
    float (*const s_part)[64 * VPT] = reinterpret_cast<float (*)[64 * VPT]>(s_hcnt + SA_HT);
    __shared__ uint32_t s_cnt, s_nseg, s_over, s_min, s_nheavy;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t me = HS == 2 ? blockIdx.x >> 1 : blockIdx.x;      // the owner id keys are matched against
    P.coff = HS == 2 ? (int)(blockIdx.x & 1) * 64 : 0;
    if (tid == 0) { s_cnt = 0; s_nseg = 0; s_over = 0; s_nheavy = 0; }
    s_hkey[tid] = SA_EMPTY;                               // (SA_HT == SA_NT)
    s_hcnt[tid] = 0;
    __syncthreads();
    int64_t rows = P.n_host;
    if (P.n_dev) {
        rows = (int64_t)P.n_dev[0] * P.n_mul;
        if (rows > P.region_stride) rows = P.region_stride;
        if (rows < 0) rows = 0;
    }
    // Both passes read the same keys.  A thread's keys are loaded SA_KPT at a time, all loads in flight together (a pass would otherwise
    // be a chain of L2 round trips: ~15 keys per thread at config 5), over the flattened (region, position) space; a list of at most
    // SA_KPT * SA_NT entries -- the usual case -- stays in registers for the second pass.
    const int64_t total = rows * P.n_regions;
    const bool resident = total <= (int64_t)SA_KPT * SA_NT;
    int64_t kreg[SA_KPT];
    uint32_t preg[SA_KPT];
    auto load_keys = [&](int64_t c0) {
#pragma unroll
        for (int u = 0; u < SA_KPT; ++u) {
            // (clamped and unconditional: behind `if (f < total)` hipcc waits for every load at the branch's join -- s_waitcnt vmcnt(0) -- and
            //  the SA_KPT loads that were meant to be in flight together were SA_KPT dependent round trips)
            const int64_t f = c0 + tid + (int64_t)u * SA_NT;
            const int64_t fc = f < total ? f : total - 1;
            int64_t r = 0, i = fc;
            if (P.n_regions > 1) { r = (int64_t)((uint32_t)fc / (uint32_t)rows); i = fc - r * rows; }      // (total < 2^32; uniform branch)
            const uint32_t pp = (uint32_t)(r * P.region_stride + i);
            const int64_t kk = sa_key<KeyT>(P, (int64_t)pp);
            preg[u] = f < total ? pp : 0u;
            kreg[u] = f < total ? kk : -1;
        }
    };
    // ---- pass A: the distinct keys this workgroup owns, with their number of contributions
    for (int64_t c0 = 0; c0 < total; c0 += (int64_t)SA_KPT * SA_NT) {
        load_keys(c0);
#pragma unroll
        for (int u = 0; u < SA_KPT; ++u) {
            const int64_t k = kreg[u];
            if (k < 0 || k >= P.R || k == P.padding_idx || sa_owner<HS>((uint32_t)k) != me) continue;
            uint32_t h = sa_slot0((uint32_t)k);
            int tries = 0;
            for (; tries < SA_HT; ++tries, h = (h + 1) & (SA_HT - 1)) {
                const uint32_t old = atomicCAS(&s_hkey[h], SA_EMPTY, (uint32_t)k);
                if (old == SA_EMPTY || old == (uint32_t)k) { atomicAdd(&s_hcnt[h], 1u); break; }
            }
            if (tries == SA_HT) s_over = 1;
        }
    }
    __syncthreads();
#ifndef SA_NO_TOUCH
    // Touch the (W, m, v) rows of the owned keys now (one thread per table slot, one word per row): against a table far beyond the TLBs' reach
    // every row is a page walk + a cold line, and they are on their way while pass B, the ordering and the run detection go on.  The values are
    // not used (kept live to the end of the function so that nothing waits for them here).
    float touch0 = 0.f, touch1 = 0.f, touch2 = 0.f;
    {
        const uint32_t hk = s_hkey[tid];
        const int64_t o = (int64_t)(hk != SA_EMPTY ? hk : 0u) * P.stride + P.coff;     // (unconditional: row 0 where the slot is empty)
        touch0 = P.W[o]; touch1 = P.m[o]; touch2 = P.v[o];
    }
#endif
    // ---- pass B: the entries of the keys with few contributions go to the list (key, position)
    if (!s_over) {
        for (int64_t c0 = 0; c0 < total; c0 += (int64_t)SA_KPT * SA_NT) {
            if (!resident) load_keys(c0);
#pragma unroll
            for (int u = 0; u < SA_KPT; ++u) {
                const int64_t k = kreg[u];
                if (k < 0 || k >= P.R || k == P.padding_idx || sa_owner<HS>((uint32_t)k) != me) continue;
                uint32_t h = sa_slot0((uint32_t)k);
                while (s_hkey[h] != (uint32_t)k) h = (h + 1) & (SA_HT - 1);
                if (s_hcnt[h] > SA_LONG) continue;
                const uint32_t at = atomicAdd(&s_cnt, 1u);
                if (at < SA_CAP) s_list[at] = ((unsigned long long)(uint32_t)k << 32) | (unsigned long long)preg[u];
            }
        }
    }
    __syncthreads();
    const uint32_t cnt = s_cnt;
    uint32_t* queue = reinterpret_cast<uint32_t*>(s_lists);
    if (s_over || cnt > SA_CAP) {
        // ---- slow exact path (more distinct keys or short-run entries than the LDS structures hold -- adversarial inputs): every owned
        //      distinct key in ascending order, each summed over the whole list
        int64_t lo = -1;
        for (;;) {
            if (tid == 0) s_min = SA_EMPTY;
            __syncthreads();
            uint32_t mine = SA_EMPTY;
            for (int r = 0; r < P.n_regions; ++r) {
                const int64_t base = (int64_t)r * P.region_stride;
                for (int64_t i = tid; i < rows; i += SA_NT) {
                    const int64_t k = sa_key<KeyT>(P, base + i);
                    if (k < 0 || k >= P.R || k == P.padding_idx || sa_owner<HS>((uint32_t)k) != me || k <= lo) continue;
                    if ((uint32_t)k < mine) mine = (uint32_t)k;
                }
            }
            if (mine != SA_EMPTY) atomicMin(&s_min, mine);
            __syncthreads();
            const uint32_t cur = s_min;
            if (cur == SA_EMPTY) break;
            sa_heavy<VPT, KeyT>(P, cur, rows, queue, s_part, lane, wave);
            lo = (int64_t)cur;
        }
        return;
    }
    // ---- order the listed entries: rank = number of smaller words (words are distinct: the position is part of them)
    for (uint32_t e = tid; e < cnt; e += SA_NT) {
        const unsigned long long x = s_list[e];
        uint32_t rank = 0;
        for (uint32_t j = 0; j < cnt; ++j) rank += (s_list[j] < x) ? 1u : 0u;
        s_sorted[rank] = x;
    }
    __syncthreads();
    // ---- runs of equal keys: one wave per run
    for (uint32_t e = tid; e < cnt; e += SA_NT)
        if (e == 0 || (uint32_t)(s_sorted[e] >> 32) != (uint32_t)(s_sorted[e - 1] >> 32)) s_seg[atomicAdd(&s_nseg, 1u)] = e;
    __syncthreads();
    const uint32_t nseg = s_nseg;
    // (SA_RUNS runs per wave at a time, every load of the batch requested before anything is used: the runs' (W, m, v) rows -- cold lines of a
    //  table far larger than the caches and the TLBs' reach -- AND the first four contribution rows of every run (most runs have one or two).
    //  Run by run, contribution loads behind the previous run's stores, a wave paid a memory round trip per run: 21 us of the launch at config 5's
    //  shapes with ALL keys distinct.  The order of every sum is unchanged.)
    for (uint32_t s0 = wave; s0 < nseg; s0 += SA_RUNS * SA_NW) {
        uint32_t key[SA_RUNS], start[SA_RUNS], len[SA_RUNS];
        SaRow<VPT> w[SA_RUNS], mm[SA_RUNS], vv[SA_RUNS], g4[SA_RUNS][4];
#pragma unroll
        for (int u = 0; u < SA_RUNS; ++u) {
            const uint32_t s = s0 + u * SA_NW;
            len[u] = 0; key[u] = 0; start[u] = 0;
            if (s < nseg) {
                start[u] = s_seg[s];
                key[u] = (uint32_t)(s_sorted[start[u]] >> 32);
                const uint32_t e = start[u] + lane;              // (a listed run has at most SA_LONG <= 64 entries)
                const bool same = e < cnt && (uint32_t)(s_sorted[e] >> 32) == key[u];
                len[u] = (uint32_t)__builtin_ctzll(~__ballot(same));
                w[u] = sa_load<VPT>(P.W, (int64_t)key[u], lane, P.stride, P.coff);
                mm[u] = sa_load<VPT>(P.m, (int64_t)key[u], lane, P.stride, P.coff);
                vv[u] = sa_load<VPT>(P.v, (int64_t)key[u], lane, P.stride, P.coff);
#pragma unroll
                for (int j = 0; j < 4; ++j) {                    // (clamped to the run's last entry: a valid row, not used)
                    const uint32_t ej = start[u] + ((uint32_t)j < len[u] ? (uint32_t)j : len[u] - 1);
                    g4[u][j] = sa_load<VPT>(P.g, (int64_t)(uint32_t)s_sorted[ej], lane, P.stride, P.coff);
                }
            }
        }
        const float ss = P.hyper ? P.hyper[0] : P.step_size, ib = P.hyper ? P.hyper[1] : P.inv_sqrt_bc2;
#pragma unroll
        for (int u = 0; u < SA_RUNS; ++u) {
            if (len[u] == 0) continue;
            SaRow<VPT> G;
            if (len[u] >= 4) {                                   // (sa_sum's order: groups of four with explicit roundings, then singles)
#pragma unroll
                for (int c = 0; c < VPT; ++c) G.x[c] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(0.f, g4[u][0].x[c]), g4[u][1].x[c]), g4[u][2].x[c]), g4[u][3].x[c]);
                if (len[u] > 4) G = sa_sum<VPT>(P.g, s_sorted, (int)start[u] + 4, (int)(start[u] + len[u]), lane, P.stride, P.coff, G);
            } else {
#pragma unroll
                for (int c = 0; c < VPT; ++c) {
                    G.x[c] = 0.f;
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        if ((uint32_t)j < len[u]) G.x[c] += g4[u][j].x[c];
                }
            }
#pragma unroll
            for (int c = 0; c < VPT; ++c) sa_adam_one(P, ss, ib, G.x[c], w[u].x[c], mm[u].x[c], vv[u].x[c]);
            sa_store<VPT>(P.W, (int64_t)key[u], lane, w[u], P.stride, P.coff);
            sa_store<VPT>(P.m, (int64_t)key[u], lane, mm[u], P.stride, P.coff);
            sa_store<VPT>(P.v, (int64_t)key[u], lane, vv[u], P.stride, P.coff);
        }
    }
    __syncthreads();
    // ---- keys with many contributions (a Zipf-head item): summed over the key list by the whole workgroup, one after the other
    if (s_hkey[tid] != SA_EMPTY && s_hcnt[tid] > SA_LONG) s_seg[atomicAdd(&s_nheavy, 1u)] = s_hkey[tid];     // (s_seg is free again)
    __syncthreads();
    const uint32_t nheavy = s_nheavy;
    for (uint32_t h = 0; h < nheavy; ++h) sa_heavy<VPT, KeyT>(P, s_seg[h], rows, queue, s_part, lane, wave);
#ifndef SA_NO_TOUCH
    asm volatile("" ::"v"(touch0), "v"(touch1), "v"(touch2));
#endif
}

