// K9: DeepFM multi-field embedding bag + FM second-order term + logistic-regression term, and BCE-with-logits.
//
// Reference (DeepFM/main.py): per field f an nn.Embedding(count_f, D=10) and an nn.Embedding(count_f, 1) (:46-54,:140-149);
//   E[b,f,:] = emb_f[x[b,f]]                      (:204-206, 2F tiny gather kernels + cat in aten)
//   lr[b]    = sum_f emb_lr_f[x[b,f]] + bias      (:58-62)
//   fm[b]    = 0.5 * sum_d ((sum_f E)^2 - sum_f E^2)   (:80-85)
//   logit    = lr + fm + dnn(flatten E);  loss = BCELoss4Logits(mean)(logit, label)   (:208-215)
// Layout: the F tables are concatenated into ONE [sum count_f, D] table (and one [sum count_f] LR vector); a field's
// ids are offset by the field's first row, so the whole multi-table lookup is one kernel and its gradient ONE
// scatter-add.  Rows are 40 B (D=10): sub-cache-line, the tables (Frappe: 5.4 k rows = 216 KB) live in L2.
// One thread per (row b, field f) reads its 10-float row with 8-byte loads; the per-row sums over fields are
// reduced across the F lanes of a row group with shuffles (F <= 64).
//
// Algorithmic bytes per row: F * (8 + 4D + 4) gathered + 4*F*D written.
#include <math.h>

#include "re_common.h"

#define FB_MAXD 16
// The sixteen words of a clamped, unconditional load batch, pinned: without a use that does not depend on the range check hipcc sinks every load
// into a branch of its own (load, s_waitcnt vmcnt(0), select -- sixteen dependent round trips).
#define FB_PIN16(t)                                                                                                                         \
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), \
                      "+v"(t[10]), "+v"(t[11]), "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]))

// lanes [0, F) of a group of FP lanes (power of two >= F) handle the F fields of one batch row
template <int FP>
__global__ __launch_bounds__(256) void fm_bag_fwd_k(const float* __restrict__ T, const float* __restrict__ TL,
                                                    const float* __restrict__ lr_bias, const int64_t* __restrict__ offsets,
                                                    int64_t rows_total, const int64_t* __restrict__ x, int64_t B, int F, int D,
                                                    float* __restrict__ E, float* __restrict__ fm_lr, int64_t* __restrict__ rows_out,
                                                    int32_t* __restrict__ keys_t) {
    const int f = threadIdx.x % FP;
    const int64_t b = (int64_t)blockIdx.x * (256 / FP) + threadIdx.x / FP;
    float e[FB_MAXD];
    float lr = 0.f;
    const bool act = b < B && f < F;
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) e[d] = 0.f;
    if (act) {
        const int64_t r = x[b * F + f] + offsets[f];
        if (rows_out) rows_out[b * F + f] = r;
        if (keys_t) keys_t[(int64_t)f * B + b] = (int32_t)x[b * F + f];     // field-major, relative to the field: re_fm_table_grad's keys
        // (one batch of unconditional loads through a clamped row / column: behind the range check and a run-time loop bound they were D + 1
        //  dependent round trips)
        const bool in = r >= 0 && r < rows_total;
        const int64_t rc = in ? r : 0;
        float t[FB_MAXD];
#pragma unroll
        for (int d = 0; d < FB_MAXD; ++d) t[d] = T[rc * D + (d < D ? d : D - 1)];
        float tl = TL[rc];
        FB_PIN16(t);
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int d = 0; d < FB_MAXD; ++d) e[d] = (in && d < D) ? t[d] : 0.f;
        lr = in ? tl : 0.f;
        float* dst = E + (b * F + f) * D;
        for (int d = 0; d < D; ++d) dst[d] = e[d];
    }
    // fm = 0.5 * sum_d ((sum_f e)^2 - sum_f e^2)
    float fm = 0.f;
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) {
        float s = e[d], q = e[d] * e[d];
#pragma unroll
        for (int o = FP / 2; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
        fm += s * s - q;
    }
#pragma unroll
    for (int o = FP / 2; o > 0; o >>= 1) lr += __shfl_xor(lr, o, 64);
    if (act && f == 0) fm_lr[b] = 0.5f * fm + lr + lr_bias[0];
}

// gE[b,f,:] = dE_mlp[b,f,:] + dlogit[b] * (sum_f' E[b,f',:] - E[b,f,:]);  gL[b,f] = dlogit[b]
template <int FP>
__global__ __launch_bounds__(256) void fm_bag_bwd_k(const float* __restrict__ E, const float* __restrict__ dE_mlp,
                                                    const float* __restrict__ dlogit, int64_t B, int F, int D,
                                                    float* __restrict__ gE, float* __restrict__ gL) {
    const int f = threadIdx.x % FP;
    const int64_t b = (int64_t)blockIdx.x * (256 / FP) + threadIdx.x / FP;
    const bool act = b < B && f < F;
    float e[FB_MAXD];
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) e[d] = 0.f;
    {
        const int64_t bc = act ? b * F + f : 0;     // (unconditional, clamped: see fm_bag_fwd_k)
        float t[FB_MAXD];
#pragma unroll
        for (int d = 0; d < FB_MAXD; ++d) t[d] = E[bc * D + (d < D ? d : D - 1)];
        FB_PIN16(t);
#pragma unroll
        for (int d = 0; d < FB_MAXD; ++d) e[d] = (act && d < D) ? t[d] : 0.f;
    }
    const float dl = (b < B) ? dlogit[b] : 0.f;
    float dm[FB_MAXD];
    {
        const float* dp = dE_mlp ? dE_mlp : E;      // (uniform; E: a valid address, the value is not used)
        const int64_t bc = act ? b * F + f : 0;
#pragma unroll
        for (int d = 0; d < FB_MAXD; ++d) dm[d] = dp[bc * D + (d < D ? d : D - 1)];
        FB_PIN16(dm);
#pragma unroll
        for (int d = 0; d < FB_MAXD; ++d) dm[d] = dE_mlp ? dm[d] : 0.f;
    }
#pragma unroll
    for (int d = 0; d < FB_MAXD; ++d) {
        float s = e[d];
#pragma unroll
        for (int o = FP / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (act && d < D) gE[(b * F + f) * D + d] = dm[d] + dl * (s - e[d]);
    }
    if (act) gL[b * F + f] = dl;
}

extern "C" int re_fm_bag_fwd(const float* T, const float* TL, const float* lr_bias, const int64_t* offsets, int64_t rows_total,
                             const int64_t* x, int64_t B, int64_t F, int64_t D, float* E, float* fm_lr, int64_t* rows_out, int32_t* keys_t,
                             re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!T || !TL || !lr_bias || !offsets || !x || !E || !fm_lr || B < 0 || rows_total <= 0) return RE_EINVAL;
    if (F < 1 || F > 64 || D < 1 || D > FB_MAXD) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define FB_FWD(FPV) hipLaunchKernelGGL(fm_bag_fwd_k<FPV>, dim3((unsigned)re_cdiv(B, 256 / FPV)), dim3(256), 0, s, T, TL, lr_bias, offsets, rows_total, x, B, (int)F, (int)D, E, fm_lr, rows_out, keys_t)
    if (F <= 8) FB_FWD(8); else if (F <= 16) FB_FWD(16); else if (F <= 32) FB_FWD(32); else FB_FWD(64);
#undef FB_FWD
    return re_launch_status();
}

extern "C" int re_fm_bag_bwd(const float* E, const float* dE_mlp, const float* dlogit, int64_t B, int64_t F, int64_t D, float* gE,
                             float* gL, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    if (!E || !dlogit || !gE || !gL || B < 0) return RE_EINVAL;
    if (F < 1 || F > 64 || D < 1 || D > FB_MAXD) return RE_EUNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
#define FB_BWD(FPV) hipLaunchKernelGGL(fm_bag_bwd_k<FPV>, dim3((unsigned)re_cdiv(B, 256 / FPV)), dim3(256), 0, s, E, dE_mlp, dlogit, B, (int)F, (int)D, gE, gL)
    if (F <= 8) FB_BWD(8); else if (F <= 16) FB_BWD(16); else if (F <= 32) FB_BWD(32); else FB_BWD(64);
#undef FB_BWD
    return re_launch_status();
}

// ---- gradients of the two concatenated tables (T [rows, D], TL [rows]) from the contribution rows of re_fm_bag_bwd, ONE launch.
// A field's B keys fall into the field's own row range, so the work is cut by ROWS: a workgroup owns a slice [row_lo, row_hi) of one field
// (slices sized by the caller for ~40 keys each when keys are uniform: ops.fm_table_slices), scans the field's B keys for the ones that fall
// into it -- waves over contiguous stretches of the batch, ballots, no barrier: the list comes out in batch order -- and sums them per row:
//   * up to 64 matches (the normal case): ONE wave sorts the (row, b) words in registers (a bitonic network of shuffles) and adds runs of equal
//     rows with a segmented scan, column by column;
//   * more (a field of a handful of values: B / count matches on a single row; or skewed keys): the list is sorted in LDS if the slice has
//     more than one row, cut into FT_GROUPS stretches walked by groups of 16 lanes (lane c < D: column c of gE, lane D: gL; sixteen loads in flight),
//     runs inside a stretch written directly, the stretches' first and last runs joined in order (16 + FT_GROUPS / 16 steps).
// Every sum's order is a fixed function of the keys: bitwise reproducible.  Rows nobody refers to are NOT written (the caller zero-fills).
// The general sorted scatter-add (scatter.hip: re_scatter_plan + two re_scatter_apply) is twelve launches of 5 - 11 us for these 40 960 keys:
// 80 of the DeepFM step's 375 us (profiles/r5_config4_kstats.txt).
#define FT_NT 512
#define FT_NW (FT_NT / 64)
#define FT_MAXB 8192
#define FT_BBITS 13
#define FT_GROUPS (FT_NT / 16)
#define FT_NONE 0xFFFFFFFFu
#define FT_INF 16          // contribution rows a lane group keeps in flight
__global__ __launch_bounds__(FT_NT) void fm_table_grad_k(const int32_t* __restrict__ keys_t, int B, int F, const int64_t* __restrict__ offsets,
                                                         int64_t rows_total, const int32_t* __restrict__ slices, const float* __restrict__ gE,
                                                         const float* __restrict__ gL, int D, float* __restrict__ gT, float* __restrict__ gTL, int P2) {
    extern __shared__ unsigned ft_lds[];
    const int Bp = (B + FT_NW * 64 - 1) / (FT_NW * 64) * (FT_NW * 64);    // a wave's stretch of the batch: Bp / 16 keys, a multiple of 64
    unsigned* const k = ft_lds;                                            // [Bp]: the waves' lists | [P2 = 2^n >= Bp]: closed up, padded, sorted
    float* const firstv = reinterpret_cast<float*>(k + Bp + P2);           // [FT_GROUPS][16]
    float* const lastv = firstv + FT_GROUPS * 16;                          // [64][16]
    unsigned* const frow = reinterpret_cast<unsigned*>(lastv + FT_GROUPS * 16);   // [64]  first run's row (FT_NONE: empty stretch)
    unsigned* const lrow = frow + FT_GROUPS;                               // [64]  last run's row (FT_NONE: the stretch is one run)
    int* const wcnt = reinterpret_cast<int*>(lrow + FT_GROUPS);            // [16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int f = slices[4 * blockIdx.x], row_lo = slices[4 * blockIdx.x + 1], row_hi = slices[4 * blockIdx.x + 2];
    const int64_t base = offsets[f];
    auto put = [&](unsigned r, int c, float v) {      // row r of this field is complete
        if (c < D) gT[(base + r) * D + c] = v;
        else if (c == D) gTL[base + r] = v;
    };
    auto val = [&](unsigned key, int c) {             // column c of the contribution row a list word names (c > D: lane D's word again, unused)
        // (ONE unconditional load through a selected pointer: behind a branch on c hipcc waits for every load on the spot -- s_waitcnt
        //  vmcnt(0) at the join -- and the sixteen loads of a pass became sixteen memory round trips)
        const int64_t cb = (int64_t)(key & (FT_MAXB - 1)) * F + f;
        const float* p = c < D ? gE + cb * D + c : gL + cb;
        return *p;
    };
    // ---- 1. the slice's keys, in batch order
    const int per_w = Bp / FT_NW;
    int n = 0;                                        // (wave-uniform)
    {
        int32_t kr[4];
        const int32_t* __restrict__ kf = keys_t + (int64_t)f * B;             // (field-major: the field's B keys are 4 B of contiguous words)
        for (int i0 = 0; i0 < per_w; i0 += 256) {     // (four loads in flight)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = wave * per_w + i0 + 64 * u + lane;
                kr[u] = kf[b < B ? b : B - 1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = wave * per_w + i0 + 64 * u + lane;
                const int64_t r = kr[u];
                const bool hit = i0 + 64 * u < per_w && b < B && r >= row_lo && r < row_hi && r < 0x7FFFF && base + r < rows_total;
                const unsigned long long hm = __ballot(hit);
                if (hm != 0ull) {
                    if (hit) k[wave * per_w + n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u))] =
                                 ((unsigned)r << FT_BBITS) | (unsigned)b;
                    n += __popcll(hm);
                }
            }
        }
    }
    if (lane == 0) wcnt[wave] = n;
    __syncthreads();
    int M = 0, below = 0;
#pragma unroll
    for (int w = 0; w < FT_NW; ++w) {
        const int x = wcnt[w];
        M += x;
        below += w < wave ? x : 0;
    }
    if (M == 0) return;
    for (int i = lane; i < n; i += 64) k[Bp + below + i] = k[wave * per_w + i];     // (closed up into the second half)
    __syncthreads();
    unsigned* const kl = k + Bp;
    const bool one_row = row_hi - row_lo == 1;
    // ---- 2a. up to 64 matches: one wave, registers
    if (M <= 64) {
        if (wave != 0) return;
        // sorted position of the lane's word = how many of the words are smaller (they are all different: 64 broadcasts and compares, no
        // dependent memory or LDS latency -- the bitonic network this replaces was 21 dependent cross-lane steps, 3 k cycles of a 5 k cycle job);
        // then ONE cross-lane move puts the words in order
        const unsigned mine = lane < M ? kl[lane] : FT_NONE;
        unsigned key = mine;
        if (!one_row) {
            int rank = 0;
            for (int j = 0; j < M; ++j) rank += ((unsigned)__builtin_amdgcn_readlane((int)mine, j) < mine) ? 1 : 0;
            // lane `rank` takes this lane's word: the inverse permutation through LDS (the list's first half is free by now)
            if (lane < M) k[rank] = mine;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            key = lane < M ? k[lane] : FT_NONE;
        }
        const unsigned r = key >> FT_BBITS;                                   // (FT_NONE >> 13: the padding's "row", behind every real one)
        const unsigned rn = (unsigned)__shfl_down((int)r, 1, 64);
        const bool run_end = lane < M && (lane == 63 || rn != r);
        // (every column requested first: a column at a time was eleven memory round trips)
        float v[16];
        const unsigned ka = lane < M ? key : kl[0];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = val(ka, c);
        bool same[6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const unsigned rt = (unsigned)__shfl_up((int)r, 1 << j, 64);      // (by every lane: a shuffle behind `lane >= ..&&` reads switched-off lanes)
            same[j] = lane >= (1 << j) && rt == r;
        }
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = lane < M ? v[c] : 0.f;
        // (the columns' scans step by step side by side: sixteen independent cross-lane moves a step instead of a chain of six per column)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float t[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = __shfl_up(v[c], 1 << j, 64);
#pragma unroll
            for (int c = 0; c < 16; ++c) v[c] += same[j] ? t[c] : 0.f;
        }
        if (run_end) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (c <= D) put(r, c, v[c]);
        }
        return;
    }
    // ---- 2b. the general form
    int Mp = FT_GROUPS * 8;
    while (Mp < M) Mp <<= 1;
    for (int i = M + tid; i < Mp; i += FT_NT) kl[i] = FT_NONE;                 // (Mp <= P2)
    if (!one_row) {
        for (int size = 2; size <= Mp; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                __syncthreads();
                for (int i = tid; i < (Mp >> 1); i += FT_NT) {
                    const int lo = 2 * i - (i & (stride - 1));
                    const unsigned a = kl[lo], b = kl[lo + stride];
                    if ((a > b) == ((lo & size) == 0)) { kl[lo] = b; kl[lo + stride] = a; }
                }
            }
    }
    __syncthreads();
    const int grp = tid >> 4, c = tid & 15;
    const int per = Mp / FT_GROUPS, e0 = grp * per;
    {
        unsigned cur = FT_NONE;
        int nrun = 0;
        float acc = 0.f;
        bool ended = false;                    // (the 16 lanes of a group agree on all of these)
        for (int eb = 0; eb < per && !ended; eb += FT_INF) {
            unsigned key[FT_INF];
            float v[FT_INF];
#pragma unroll
            for (int u = 0; u < FT_INF; ++u) {
                key[u] = eb + u < per ? kl[e0 + eb + u] : FT_NONE;
                v[u] = val(key[u] == FT_NONE ? kl[0] : key[u], c);            // (a valid address: the loads are unconditional)
            }
#pragma unroll
            for (int u = 0; u < FT_INF; ++u) {
                if (key[u] == FT_NONE) { ended = true; break; }               // (padding sorts to the end)
                const unsigned r = key[u] >> FT_BBITS;
                if (r != cur) {
                    if (cur != FT_NONE) {
                        if (nrun == 0) { firstv[grp * 16 + c] = acc; if (c == 0) frow[grp] = cur; }
                        else put(cur, c, acc);
                        ++nrun;
                    }
                    cur = r;
                    acc = 0.f;
                }
                acc += v[u];
            }
        }
        if (cur == FT_NONE) { if (c == 0) { frow[grp] = FT_NONE; lrow[grp] = FT_NONE; } }
        else if (nrun == 0) { firstv[grp * 16 + c] = acc; if (c == 0) { frow[grp] = cur; lrow[grp] = FT_NONE; } }
        else { lastv[grp * 16 + c] = acc; if (c == 0) lrow[grp] = cur; }
    }
    __syncthreads();
    // the stretches' open ends, in order: FT_GROUPS / 16 lane groups join sixteen stretches each, then one joins those
    auto join = [&](int g0, unsigned& hr, float& hv, bool& hopen, unsigned& cr, float& carry) {
        // walks stretches [g0, g0 + 16): (hr, hv) = the FIRST run met (hopen: nothing has closed it yet), (cr, carry) = the run still open at the
        // end.  (The sixteen stretches' words come out of LDS together: fetched one at a time they were four dependent LDS trips a step.)
        unsigned fr[16], lr[16];
        float fv[16], lv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { fr[i] = frow[g0 + i]; lr[i] = lrow[g0 + i]; fv[i] = firstv[(g0 + i) * 16 + c]; lv[i] = lastv[(g0 + i) * 16 + c]; }
        hr = FT_NONE; hv = 0.f; hopen = true; cr = FT_NONE; carry = 0.f;
        bool done = false;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (fr[i] == FT_NONE) done = true;
            if (!done) {
                if (fr[i] == cr) carry += fv[i];
                else {
                    if (cr != FT_NONE) { if (hopen && hr == cr) { hv = carry; hopen = false; } else put(cr, c, carry); }
                    cr = fr[i]; carry = fv[i];
                    if (hr == FT_NONE) hr = fr[i];
                }
                if (lr[i] != FT_NONE) {
                    if (hopen && hr == cr) { hv = carry; hopen = false; } else put(cr, c, carry);
                    cr = lr[i]; carry = lv[i];
                }
            }
        }
    };
    constexpr int FT_NQ = FT_GROUPS / 16;
    if (tid < 16 * FT_NQ) {
        unsigned hr, cr;
        float hv, carry;
        bool hopen;
        join(grp * 16, hr, hv, hopen, cr, carry);
        // a quarter's summary in the slots of its first stretch: first run (row, value; still open = the quarter is ONE run) and last run
        // (the four quarters are lanes of one wave: its LDS reads above are done before these writes)
        if (hr == FT_NONE) { if (c == 0) { frow[grp * 16] = FT_NONE; lrow[grp * 16] = FT_NONE; } }
        else if (hopen) { firstv[grp * 256 + c] = carry; if (c == 0) { frow[grp * 16] = hr; lrow[grp * 16] = FT_NONE; } }
        else { firstv[grp * 256 + c] = hv; lastv[grp * 256 + c] = carry; if (c == 0) { frow[grp * 16] = hr; lrow[grp * 16] = cr; } }
    }
    __syncthreads();
    if (tid < 16) {
        unsigned cr = FT_NONE;
        float carry = 0.f;
        for (int q = 0; q < FT_NQ; ++q) {
            const unsigned fr = frow[q * 16];
            if (fr == FT_NONE) break;
            const float fv = firstv[q * 256 + c];
            if (fr == cr) carry += fv;
            else { if (cr != FT_NONE) put(cr, c, carry); cr = fr; carry = fv; }
            const unsigned lr = lrow[q * 16];
            if (lr != FT_NONE) { put(cr, c, carry); cr = lr; carry = lastv[q * 256 + c]; }
        }
        if (cr != FT_NONE) put(cr, c, carry);
    }
}

extern "C" int re_fm_table_grad(const int32_t* keys_t, int64_t B, int64_t F, const int64_t* offsets, int64_t rows_total, const int32_t* slices,
                                int64_t n_slices, const float* gE, const float* gL, int64_t D, float* gT, float* gTL, re_stream_t stream) {
    re_clear_error();
    if (B == 0 || n_slices == 0) return RE_OK;
    if (!keys_t || !offsets || !slices || !gE || !gL || !gT || !gTL || B < 0 || rows_total <= 0 || n_slices < 0) return RE_EINVAL;
    if (F < 1 || F > 64 || D < 1 || D > 15 || B > FT_MAXB || n_slices > (1 << 20)) return RE_EUNSUPPORTED;
    const int64_t Bp = re_cdiv(B, FT_NT) * FT_NT;
    int P2 = FT_GROUPS * 8;
    while (P2 < Bp) P2 <<= 1;
    const size_t ldsb = (size_t)(Bp + P2) * 4 + 2 * FT_GROUPS * 16 * 4 + 2 * FT_GROUPS * 4 + FT_NW * 4;
    if (hipFuncSetAttribute((const void*)fm_table_grad_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
    hipLaunchKernelGGL(fm_table_grad_k, dim3((unsigned)n_slices), dim3(FT_NT), ldsb, (hipStream_t)stream, keys_t, (int)B, (int)F, offsets, rows_total,
                       slices, gE, gL, (int)D, gT, gTL, P2);
    return re_launch_status();
}

// BCE with logits, reduction = mean (DeepFM/main.py:214): loss = mean(max(x,0) - x*y + log1p(exp(-|x|)));
// dlogit = (sigmoid(x) - y) / n;  dsum[0] = sum(dlogit) (gradient of the LR bias).  One block, fixed-order sums.
__global__ __launch_bounds__(1024) void bce_logits_k(const float* __restrict__ x, const float* __restrict__ y, int64_t n,
                                                     float* __restrict__ loss, float* __restrict__ dlogit, float* __restrict__ dsum) {
    __shared__ float s1[16], s2[16];
    float l = 0.f, g = 0.f;
    const float inv = 1.0f / (float)n;
    for (int64_t i = threadIdx.x; i < n; i += 1024) {
        const float xi = x[i], yi = y[i];
        l += fmaxf(xi, 0.f) - xi * yi + log1pf(expf(-fabsf(xi)));
        const float d = (re_sigmoid(xi) - yi) * inv;
        dlogit[i] = d;
        g += d;
    }
    l = re_wave_sum(l);
    g = re_wave_sum(g);
    if ((threadIdx.x & 63) == 0) { s1[threadIdx.x >> 6] = l; s2[threadIdx.x >> 6] = g; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 16; ++w) { a += s1[w]; b += s2[w]; }
        loss[0] = a * inv;
        if (dsum) dsum[0] = b;
    }
}

extern "C" int re_bce_logits(const float* logits, const float* labels, int64_t n, float* loss, float* dlogit, float* dsum,
                             re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !loss || !dlogit || n <= 0) return RE_EINVAL;
    hipLaunchKernelGGL(bce_logits_k, dim3(1), dim3(1024), 0, (hipStream_t)stream, logits, labels, n, loss, dlogit, dsum);
    return re_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// Cross entropy over materialised logits, forward + backward in place (SASRec --loss CE, SASRec/main.py:217-219;
// CrossEntropy4Logits(reduction="mean") = F.cross_entropy).  One workgroup per row: max, sum-exp, then
//   row_loss[m] = logsumexp(x_m) - x_m[y_m];   x_m <- (softmax(x_m) - onehot(y_m)) * (1 / M)      (the gradient)
// loss[0] = mean(row_loss) by a second, fixed-order kernel.  [M, N] = 3 000 x 12 101 on the benchmark shapes: 145 MB,
// read twice and written once.
__global__ __launch_bounds__(256) void ce_rows_k(float* __restrict__ logits, int64_t N, int64_t ld, const int64_t* __restrict__ labels,
                                                 float inv_m, float* __restrict__ row_loss) {
    __shared__ float red[4];
    float* x = logits + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int64_t i = tid; i < N; i += 256) mx = fmaxf(mx, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int64_t i = tid; i < N; i += 256) s += expf(x[i] - mx);
    s = re_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    s = ((red[0] + red[1]) + red[2]) + red[3];
    const int64_t y = labels[blockIdx.x];
    const bool yok = y >= 0 && y < N;
    if (tid == 0) row_loss[blockIdx.x] = yok ? (logf(s) + mx - x[y]) : 0.f;
    __syncthreads();
    const float inv_s = 1.0f / s;
    for (int64_t i = tid; i < N; i += 256) {
        float p = expf(x[i] - mx) * inv_s;
        if (yok && i == y) p -= 1.0f;
        x[i] = p * inv_m;
    }
}

__global__ __launch_bounds__(256) void sum_mean_k(const float* __restrict__ v, int64_t n, float scale, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) s += v[i];
    s = re_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (((red[0] + red[1]) + red[2]) + red[3]) * scale;
}

extern "C" int re_ce_rows(float* logits, int64_t M, int64_t N, int64_t ld, const int64_t* labels, float* row_loss, float* loss,
                          re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !row_loss || !loss || M <= 0 || N <= 0 || ld < N) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_rows_k, dim3((unsigned)M), dim3(256), 0, s, logits, N, ld, labels, 1.0f / (float)M, row_loss);
    hipLaunchKernelGGL(sum_mean_k, dim3(1), dim3(256), 0, s, (const float*)row_loss, M, 1.0f / (float)M, loss);
    return re_launch_status();
}

// ---------------------------------------------------------------------------------------------------------
// The same cross entropy WITHOUT the [M, N] matrix: the catalog is walked in column chunks, the caller materialises one chunk of
// logits [M, Nc] at a time (one GEMM), and the row statistics are carried between chunks (online log-sum-exp):
//   pass 1, per chunk:  (rowmax, rowsum) <- merge with the chunk's (max, sum exp(x - max)); tgt[m] = x[m, y_m] when y_m is in the chunk
//   finish:             row_loss = log(rowsum) + rowmax - tgt;  loss = mean
//   pass 2, per chunk (logits recomputed by the same GEMM):  x <- (exp(x - rowmax) / rowsum - [col == y_m]) / M   in place
// Memory is M x Nc instead of M x N (a 1 M-item catalog at M = 4 000 would be 16 GB of logits); the price is one more GEMM per chunk.
__global__ __launch_bounds__(256) void ce_chunk_stats_k(const float* __restrict__ logits, int64_t Nc, int64_t ld, int64_t col0,
                                                        const int64_t* __restrict__ labels, int first, float* __restrict__ rowmax,
                                                        float* __restrict__ rowsum, float* __restrict__ tgt) {
    __shared__ float red[4];
    const float* x = logits + (int64_t)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int64_t i = tid; i < Nc; i += 256) mx = fmaxf(mx, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int64_t i = tid; i < Nc; i += 256) s += expf(x[i] - mx);
    s = re_wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        s = ((red[0] + red[1]) + red[2]) + red[3];
        if (!first) {
            const float m0 = rowmax[blockIdx.x], s0 = rowsum[blockIdx.x];
            const float m1 = fmaxf(m0, mx);
            s = s0 * expf(m0 - m1) + s * expf(mx - m1);
            mx = m1;
        }
        rowmax[blockIdx.x] = mx;
        rowsum[blockIdx.x] = s;
        const int64_t y = labels[blockIdx.x] - col0;
        if (y >= 0 && y < Nc) tgt[blockIdx.x] = x[y];
        else if (first) tgt[blockIdx.x] = 0.f;
    }
}

__global__ __launch_bounds__(256) void ce_chunk_loss_k(const float* __restrict__ rowmax, const float* __restrict__ rowsum, const float* __restrict__ tgt,
                                                       const int64_t* __restrict__ labels, int64_t M, int64_t N, float* __restrict__ row_loss) {
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
        const int64_t y = labels[m];
        row_loss[m] = (y >= 0 && y < N) ? logf(rowsum[m]) + rowmax[m] - tgt[m] : 0.f;
    }
}

__global__ __launch_bounds__(256) void ce_chunk_grad_k(float* __restrict__ logits, int64_t Nc, int64_t ld, int64_t col0,
                                                       const int64_t* __restrict__ labels, const float* __restrict__ rowmax,
                                                       const float* __restrict__ rowsum, float inv_m) {
    float* x = logits + (int64_t)blockIdx.x * ld;
    const float mx = rowmax[blockIdx.x], inv_s = 1.0f / rowsum[blockIdx.x];
    const int64_t y = labels[blockIdx.x] - col0;
    for (int64_t i = threadIdx.x; i < Nc; i += 256) {
        float p = expf(x[i] - mx) * inv_s;
        if (i == y) p -= 1.0f;
        x[i] = p * inv_m;
    }
}

extern "C" int re_ce_chunk_stats(const float* logits, int64_t M, int64_t Nc, int64_t ld, int64_t col0, const int64_t* labels, int first,
                                 float* rowmax, float* rowsum, float* tgt, re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !rowmax || !rowsum || !tgt || M <= 0 || Nc <= 0 || ld < Nc || col0 < 0) return RE_EINVAL;
    hipLaunchKernelGGL(ce_chunk_stats_k, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, logits, Nc, ld, col0, labels, first, rowmax, rowsum, tgt);
    return re_launch_status();
}

extern "C" int re_ce_chunk_loss(const float* rowmax, const float* rowsum, const float* tgt, const int64_t* labels, int64_t M, int64_t N,
                                float* row_loss, float* loss, re_stream_t stream) {
    re_clear_error();
    if (!rowmax || !rowsum || !tgt || !labels || !row_loss || !loss || M <= 0 || N <= 0) return RE_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(ce_chunk_loss_k, dim3(re_grid(M, 256)), dim3(256), 0, s, rowmax, rowsum, tgt, labels, M, N, row_loss);
    hipLaunchKernelGGL(sum_mean_k, dim3(1), dim3(256), 0, s, (const float*)row_loss, M, 1.0f / (float)M, loss);
    return re_launch_status();
}

extern "C" int re_ce_chunk_grad(float* logits, int64_t M, int64_t Nc, int64_t ld, int64_t col0, const int64_t* labels, const float* rowmax,
                                const float* rowsum, re_stream_t stream) {
    re_clear_error();
    if (!logits || !labels || !rowmax || !rowsum || M <= 0 || Nc <= 0 || ld < Nc || col0 < 0) return RE_EINVAL;
    hipLaunchKernelGGL(ce_chunk_grad_k, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, logits, Nc, ld, col0, labels, rowmax, rowsum,
                       1.0f / (float)M);
    return re_launch_status();
}
