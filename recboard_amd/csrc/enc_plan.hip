// Batch preparation of a SASRec training / evaluation step as ONE device launch (no host sync, capturable):
//   * what the reference does at the top of `fit` (SASRec/main.py:199-204): the mask of non-pad positions, their number M (the
//     denominator of the mean loss), and the destination rows of the 3*B*S item-gradient contributions
//     (seq | pos + 1 | neg + 1, 0 = padding row = dropped) for the scatter-add;
//   * the encoder kernels' work plan (enc_common.h): which (sequence, position) every compact row holds, which tiles form
//     a work item;
//   * optionally copies (seq, pos, neg) into the static buffers a captured step reads and writes the step scalars
//     { seed, 0, lr / (1 - b1^t), 1 / sqrt(1 - b2^t) }.
// Workgroup 0 builds the plan; the other workgroups do the element-wise part.
#include <math.h>

#include "enc_common.h"

#define PL_NT 1024
#define PL_NW (PL_NT / 64)
#define PL_NCLS 8   // 0..2: long sequences of 4 / 3 / 2 tiles; 3..7: slots of 16 / 8 / 4 / 2 / 1 rows

__device__ __forceinline__ int pl_class(int span) {
    if (span > 48) return 0;
    if (span > 32) return 1;
    if (span > 16) return 2;
    if (span > 8) return 3;
    if (span > 4) return 4;
    if (span > 2) return 5;
    if (span > 1) return 6;
    return 7;
}

__global__ __launch_bounds__(PL_NT) void sasrec_batch_prep_k(const int64_t* __restrict__ seq, const int64_t* __restrict__ pos,
                                                             const int64_t* __restrict__ neg, int B, int S, int ncu, int max_tiles,
                                                             int64_t* __restrict__ seq_out, int64_t* __restrict__ pos_out,
                                                             int64_t* __restrict__ neg_out, uint8_t* __restrict__ valid,
                                                             int* __restrict__ count, int64_t* __restrict__ rows_all, int* __restrict__ plan,
                                                             uint32_t* __restrict__ state, uint32_t seed, float step_size, float inv_sqrt_bc2) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x > 0) {
        // ---- element-wise part: copies, valid mask, scatter destination rows
        const int64_t n = (int64_t)B * S;
        for (int64_t i = (int64_t)(blockIdx.x - 1) * PL_NT + tid; i < n; i += (int64_t)(gridDim.x - 1) * PL_NT) {
            const int64_t s = seq[i];
            const bool v = s != 0;
            if (seq_out) seq_out[i] = s;
            if (valid) valid[i] = v ? 1 : 0;
            if (pos) {
                const int64_t p = pos[i], q = neg[i];
                if (pos_out) { pos_out[i] = p; neg_out[i] = q; }
                if (rows_all) {
                    rows_all[i] = s;
                    rows_all[n + i] = v ? p + 1 : 0;
                    rows_all[2 * n + i] = v ? q + 1 : 0;
                }
            }
        }
        return;
    }
    // ---- plan (workgroup 0)
    __shared__ int s_cnt[PL_NCLS][PL_NW];
    __shared__ int s_tot[PL_NCLS], s_base[PL_NCLS], s_lay[16], s_red[PL_NW];
    const int64_t mt = enc_plan_max_tiles(B, S);
    int* hdr = plan;
    int* items = plan + EP_HDR;
    int2* rowmap = (int2*)(plan + enc_plan_rowmap_word(B, S));
    int* sc_span = plan + enc_plan_rowmap_word(B, S) + 2 * 16 * mt;   // [B] span
    int* sc_place = sc_span + B;                                      // [B] first compact row of the sequence
    if (tid == 0 && state) {
        state[0] = seed;
        state[1] = 0u;
        state[2] = __float_as_uint(step_size);
        state[3] = __float_as_uint(inv_sqrt_bc2);
    }
    if (tid < PL_NCLS) s_tot[tid] = 0;
    // 1. span of every sequence (a wave per sequence, lane = position); a sequence without any item is given one explicit pad row
    int nnz = 0;
    for (int b = wave; b < B; b += PL_NW) {
        const bool nzl = lane < S && seq[(int64_t)b * S + lane] != 0;
        const unsigned long long m = __ballot(nzl);
        const int first = m ? __builtin_ctzll(m) : S - 1;
        if (lane == 0) {
            sc_span[b] = S - first;
            nnz += __builtin_popcountll(m);
        }
    }
    if (lane == 0) s_red[wave] = nnz;
    __syncthreads();
    if (tid == 0) {
        int c = 0;
        for (int w = 0; w < PL_NW; ++w) c += s_red[w];
        hdr[4] = c;
        if (count) count[0] = c;
    }
    // 2. class totals, then ranks (ballot prefix counts: deterministic), in chunks of PL_NT sequences
    for (int pass = 0; pass < 2; ++pass) {
        if (pass == 1) {
            __syncthreads();
            if (tid == 0) {
                const int n0 = s_tot[0], n1 = s_tot[1], n2 = s_tot[2];
                const int nlong = n0 + n1 + n2, tlong = 4 * n0 + 3 * n1 + 2 * n2;
                int ro[PL_NCLS];
                ro[3] = 0;
                ro[4] = ro[3] + 16 * s_tot[3];
                ro[5] = ro[4] + 8 * s_tot[4];
                ro[6] = ro[5] + 4 * s_tot[5];
                ro[7] = ro[6] + 2 * s_tot[6];
                const int rs = ro[7] + s_tot[7];
                const int tshort = (rs + 15) >> 4;
                int avail = ncu - nlong;
                if (avail < 1) avail = 1;
                int G = (tshort + avail - 1) / avail;
                if (G < 1) G = 1;
                if (G > max_tiles) G = max_tiles;
                const int nshort = (tshort + G - 1) / G;
                s_lay[0] = nlong; s_lay[1] = tlong; s_lay[2] = tshort; s_lay[3] = G; s_lay[4] = nshort;
                s_lay[5] = 0; s_lay[6] = 4 * n0; s_lay[7] = 4 * n0 + 3 * n1;                 // first tile of the long classes
                s_lay[8] = 0; s_lay[9] = n0; s_lay[10] = n0 + n1;                            // first item of the long classes
                for (int k = 3; k < PL_NCLS; ++k) s_lay[8 + k] = 16 * tlong + ro[k];         // first compact row of the slot classes
                hdr[0] = nlong + nshort; hdr[1] = tlong + tshort; hdr[2] = nlong; hdr[3] = G; hdr[5] = 0; hdr[6] = 0; hdr[7] = 0;
                for (int k = 0; k < PL_NCLS; ++k) s_base[k] = 0;
            }
            __syncthreads();
        }
        for (int b0 = 0; b0 < B; b0 += PL_NT) {
            const int b = b0 + tid;
            const int span = b < B ? sc_span[b] : 0;
            const int cls = b < B ? pl_class(span) : -1;
            int rank = 0;
#pragma unroll
            for (int k = 0; k < PL_NCLS; ++k) {
                const unsigned long long m = __ballot(cls == k);
                if (lane == 0) s_cnt[k][wave] = __builtin_popcountll(m);
                if (cls == k) rank = __builtin_popcountll(m & ((1ull << lane) - 1ull));
            }
            __syncthreads();
            if (pass == 1 && cls >= 0) {
                for (int w = 0; w < wave; ++w) rank += s_cnt[cls][w];
                rank += s_base[cls];
                int row0;
                if (cls < 3) {
                    const int nt = 4 - cls;
                    const int tile0 = s_lay[5 + cls] + nt * rank;
                    items[s_lay[8 + cls] + rank] = tile0 | (nt << 24) | (1 << 28);
                    row0 = 16 * tile0;
                } else {
                    row0 = s_lay[8 + cls] + (16 >> (cls - 3)) * rank;
                }
                sc_place[b] = row0;
            }
            __syncthreads();
            if (tid < PL_NCLS) {
                int c = 0;
                for (int w = 0; w < PL_NW; ++w) c += s_cnt[tid][w];
                if (pass == 0) s_tot[tid] += c; else s_base[tid] += c;
            }
            __syncthreads();
        }
    }
    // 3. short items, dummy rows, then the rows of every sequence
    const int nlong = s_lay[0], tlong = s_lay[1], tshort = s_lay[2], G = s_lay[3], nshort = s_lay[4];
    for (int i = tid; i < nshort; i += PL_NT) {
        const int t0 = i * G;
        const int nt = (tshort - t0) < G ? (tshort - t0) : G;
        items[nlong + i] = (tlong + t0) | (nt << 24);
    }
    const int nrows = 16 * (tlong + tshort);
    for (int i = tid; i < nrows; i += PL_NT) rowmap[i] = make_int2(-1, 0);
    __syncthreads();
    for (int b = wave; b < B; b += PL_NW) {
        const int span = sc_span[b], row0 = sc_place[b];
        if (lane < span) rowmap[row0 + lane] = make_int2(b * S + (S - span) + lane, S - span);
    }
}

extern "C" size_t re_sasrec_plan_bytes(int64_t B, int64_t S) {
    if (B <= 0 || S <= 0) return 256;
    return re_align(enc_plan_bytes(B, S));
}

extern "C" int re_sasrec_batch_prep(const int64_t* seq, const int64_t* pos, const int64_t* neg, int64_t B, int64_t S, int32_t ncu,
                                    int32_t max_tiles, int64_t* seq_out, int64_t* pos_out, int64_t* neg_out, uint8_t* valid,
                                    int32_t* count, int64_t* rows_all, void* plan, size_t plan_bytes, uint32_t* state, uint32_t seed,
                                    int64_t step, double lr, double beta1, double beta2, re_stream_t stream) {
    re_clear_error();
    if (!seq || !plan || B <= 0 || S <= 0) return RE_EINVAL;
    if ((pos == nullptr) != (neg == nullptr) || (pos_out == nullptr) != (neg_out == nullptr)) return RE_EINVAL;
    if (S > 64 || B * S > (int64_t)1 << 30 || max_tiles < 1 || max_tiles > 4 || 16 * max_tiles < S) return RE_EUNSUPPORTED;
    if (plan_bytes < enc_plan_bytes(B, S)) return RE_EWORKSPACE;
    if (state && step < 1) return RE_EINVAL;
    if (ncu < 1) ncu = 1;
    float ss = 0.f, ib = 0.f;
    if (state) {
        ss = (float)(lr / (1.0 - pow(beta1, (double)step)));
        ib = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
    }
    const bool elementwise = seq_out || valid || rows_all || pos_out;
    const unsigned grid = 1 + (elementwise ? re_grid(B * S, 4 * PL_NT, 64) : 0);
    hipLaunchKernelGGL(sasrec_batch_prep_k, dim3(grid), dim3(PL_NT), 0, (hipStream_t)stream, seq, pos, neg, (int)B, (int)S, (int)ncu,
                       (int)max_tiles, seq_out, pos_out, neg_out, valid, count, rows_all, (int*)plan, state, seed, ss, ib);
    return re_launch_status();
}
