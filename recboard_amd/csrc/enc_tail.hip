// The tail of the D = 64 SASRec training step as ONE launch + the reduction: the item table's scatter-add (+ its Adam) and the encoder's
// weight-gradient jobs both depend on the item kernels alone.  As two graph branches (round 3's first form) the fork and the join cost more than
// half of what the overlap saved (timeline of a captured step: the second branch started 13 us behind the item kernels, the node behind the join
// 10 us behind the branches' end).  Here the scatter-add's 1024-thread workgroups (scatter_owner.h: all but the Zipf head's owner are done in
// about half the launch) go on with the weight-gradient jobs (enc_wgrad_job.h), two at a time in their two halves, handed out by a ticket counter
// -- the workgroup with the hot row never gets to take one.  Results: those of re_scatter_adam_rows_small and re_sasrec_encoder_step_part(part =
// 4), bit for bit (a job's partial does not depend on who computes it; the reduction adds the partials in split order).
#include <hip/hip_runtime.h>
#ifdef TAIL_PROFILE
// Diagnostic build (`make encprof`; scripts/tail_phases.py): shader-clock stamps of every workgroup of the last enc_tail_k launch:
// [0] start, [1] scatter-add done, [2 + 2 i] ticket of its i-th job + 1, [3 + 2 i] that job's end (i < 3), [8 .. 13] the plan job's phases
// (in the workgroup that ran it), [15] end
#define TAIL_MARKS 32
#define TAIL_MARK_WGS 4096
__device__ unsigned long long g_tail_marks[TAIL_MARK_WGS * TAIL_MARKS];
extern "C" int re_dbg_tail_marks(unsigned long long* out, int nwg) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tail_marks), sizeof(unsigned long long) * TAIL_MARKS * (nwg < TAIL_MARK_WGS ? nwg : TAIL_MARK_WGS)) == hipSuccess ? 0 : 1;
}
#define TAIL_MARK(i, v) do { if (threadIdx.x == 0 && blockIdx.x < TAIL_MARK_WGS) g_tail_marks[blockIdx.x * TAIL_MARKS + (i)] = (v); } while (0)
#define TAIL_NOW() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define PL_STAMP(i) TAIL_MARK(8 + (i), TAIL_NOW())
#define WG_STAMP(i) TAIL_MARK(8 + (i), TAIL_NOW())
#define SO_STAMP(i) TAIL_MARK(20 + (i), TAIL_NOW())     /* scatter_owner.h: 0 start, 1 key count known, 2 keys scanned, 3 rows added, 4 rows stored */
#define TJ_STAMP(i) TAIL_MARK(16 + (i), TAIL_NOW())     /* tail_jobs, first pass: 0 first barrier passed, 1 ticket read, 2 job called */
#else
#define TAIL_MARK(i, v) do { } while (0)
#define TAIL_NOW() 0ull
#define TJ_STAMP(i) do { } while (0)
#endif
#include "enc_wgrad_job.h"
#include "scatter_owner.h"
#include "adam_rows_owner.h"
#include "enc_plan_body.h"

struct TailJobs {
    const float* tape;
    EncTape T;
    const float* gtape;
    int64_t NR;
    const void* plan;
    int B, S, L;
    float* part;
    const int64_t* seq;
    const float* contrib;
    float* ppart;        // nullptr: no position-table gradient
    int nsplit;          // row splits of a matrix's contraction (enc_wgrad_job.h: wg_nsplit_tail)
    unsigned* ticket;    // zero at launch; enc_grad_reduce_k (the next launch) zeroes it again.  (Round 5 tried the reduction as the queue's LAST jobs,
                         // behind done counts -- exact, and slower: profiles/r5_tail_reduce_in_queue.txt; that form is commit 4086ee1.)
};

// The NEXT batch's preparation (enc_plan_body.h: it depends on the batch alone) as the first jobs of the ticket queue: one plan job, n_ew
// element-wise jobs -- in front of a step it is a 14 us launch (one workgroup's chain of barriers and round trips), here it rides in
// workgroups that are done with the table while others still work.  mail: a PlMail in device memory (the next batch's tensors or its sampling
// source), written by re_sasrec_step_stage[_sample] in front of this step (neither: no next batch); outputs: the OTHER captured copy's buffers.
struct TailPrep {
    const PlMail* mail;           // nullptr: this launch prepares nothing
    int B, S, ncu, max_tiles, split_long, n_ew, span_parts;
    int64_t *seq_out, *pos_out, *neg_out;
    uint8_t* valid;
    int* count;
    int64_t* rows_all;
    int* plan;
};
static_assert(PL_NT == SO_NT && PL_NT == SA_NT, "the preparation jobs are written for the tail launches' workgroup size");

// ---- weight-gradient jobs.  D = 64: two per ticket: matrix jobs q = 2 t + half -> (block, matrix, split) = (q / 144, q / 24 % 6, q % 24); D = 128:
//      one per ticket, run by the whole workgroup: (t / 72, t / 12 % 6, t % 12).  Then the position-table jobs (WG_POS_GROUPS ranges of the batch's
//      sequences, two per ticket) -- both halves of a workgroup always run the same kind
__shared__ int s_job;   // the workgroup's current ticket

// The queue's counter is one address behind an agent-scope atomic: the first ticket's round trip (and the other 255 workgroups' turns at the
// counter) runs under this workgroup's last contribution-row loads (scatter_owner.h: issue / collect) -- unless the workgroup owns a hot row
// (several times its share of the matches): that one comes to the queue when it is done, as before, and normally finds it empty.
struct TailEarly {
    unsigned* ticket;
    int nwg;
    bool has = false;   // (uniform)
    int val = -1;       // (thread 0)
    __device__ __forceinline__ void issue(int matches, int nkeys) {
        has = matches <= 4 * (nkeys / nwg) + 64;
        if (has && threadIdx.x == 0) val = (int)atomicAdd(ticket, 1u);
    }
    __device__ __forceinline__ void collect() {
        if (has && threadIdx.x == 0) s_job = val;
    }
};

// has_early: s_job already holds this workgroup's first ticket
template <int D>
__device__ __forceinline__ void tail_jobs(const TailJobs& J, const TailPrep& TP, float* lds, int n_tiles, bool has_early = false) {
    const int tid = threadIdx.x, half = tid >> 9, ht = tid & 511;
    float* jl = lds + half * wg_job_lds_floats<D>();
    const int WG_NSPLIT = J.nsplit;
    const int PER_PLANE = WG_NSPLIT * EG_NMAT;          // matrix jobs of a block
    constexpr bool WHOLE = D == 128;                    // a matrix job is run by the whole workgroup (enc_wgrad_job.h: wg_nsplit), not two by its halves
    constexpr int POS_GROUPS = WG_POS_GROUPS;           // the position jobs: one per 512-thread group (two per ticket)
    const int n_mat = WHOLE ? J.L * PER_PLANE : J.L * PER_PLANE / 2, n_pos = J.ppart ? POS_GROUPS / 2 : 0;
    const int n_prep = TP.mail ? 1 + TP.n_ew : 0;
    int jn = 0;
    for (;;) {
        if (jn == 0) TJ_STAMP(3);
        __syncthreads();   // (the launch's first part / the previous job's stages are done with the LDS)
        if (jn > 0 && jn <= 3) TAIL_MARK(1 + 2 * jn, TAIL_NOW());
        if (jn == 0) TJ_STAMP(0);
        if (tid == 0 && !(has_early && jn == 0)) s_job = (int)atomicAdd(J.ticket, 1u);
        __syncthreads();
        int t = s_job;
        if (jn == 0) TJ_STAMP(1);
        if (t >= n_prep + n_mat + n_pos) break;
        if (jn < 3) TAIL_MARK(2 + 2 * jn, (unsigned long long)(t + 1));
        ++jn;
        // Queue order: the plan job (the longest single job), the matrix tickets, then the short ones -- the next batch's element-wise jobs and the
        // position tickets -- for the workgroups that come late (owners of hot rows) or have finished a first ticket: with the element-wise
        // jobs in front, a batch of 4 096 sequences had 200 workgroups spend their first 15 k cycles on them and start the 110 k-cycle matrix
        // tickets behind (scripts/tail_phases.py).
        const int n_plan = n_prep ? 1 : 0;
        if (t < n_plan || (t >= n_plan + n_mat && t < n_prep + n_mat)) {
            const PlMail M = *TP.mail;
            if (M.seq || M.SP.ptr) {                           // (uniform)
                if (t == 0) pl_plan(M.seq, TP.B, TP.S, TP.ncu, TP.max_tiles, TP.split_long, TP.count, TP.plan, M.SP, reinterpret_cast<unsigned char*>(lds), PL_MODE_REST, M.epoch);
                else pl_elementwise(t - n_plan - n_mat, TP.n_ew, M.seq, M.pos, M.neg, TP.B, TP.S, TP.seq_out, TP.pos_out, TP.neg_out, TP.valid, TP.rows_all, M.SP);
            }
            continue;
        }
        if (t < n_plan + n_mat) {
            t -= n_plan;
            if (jn == 1) TJ_STAMP(2);
            if constexpr (WHOLE) {
                wg_matrix_job<D, SO_NT>(tid, lds, t / PER_PLANE, (t / WG_NSPLIT) % EG_NMAT, t % WG_NSPLIT, J.tape, J.T, J.gtape, J.NR, n_tiles, J.part, WG_NSPLIT);
            } else {
                const int q = 2 * t + half;
                wg_matrix_job<D>(ht, jl, q / PER_PLANE, (q / WG_NSPLIT) % EG_NMAT, q % WG_NSPLIT, J.tape, J.T, J.gtape, J.NR, n_tiles, J.part, WG_NSPLIT);
            }
        } else {
            wg_pos_job<D>(ht, jl, 2 * (t - n_prep - n_mat) + half, J.B, J.S, J.seq, J.contrib, J.ppart);
        }
    }
}

// The next batch's spans (the plan's phase 1: the only part that reads the batch) by the launch's LAST workgroup, from the launch's start:
// the plan job of the queue (PL_MODE_REST) then starts ~23 k cycles in with the spans already there, and ends with the weight-gradient tickets
// instead of ~17 k cycles behind them (scripts/tail_phases.py).  The last workgroup owns cold rows (popular items have small ids) and, coming
// late to the queue, takes no ticket: its extra work is hidden.
__device__ __forceinline__ void tail_spans(const TailPrep& TP, float* lds) {
    const int parts = TP.span_parts;   // (the launch's LAST workgroups: one up to 512 sequences, one more per 512 from there on, sixteen at most)
    if (!TP.mail || (int)blockIdx.x < (int)gridDim.x - parts) return;
    const PlMail M = *TP.mail;
    if (M.seq || M.SP.ptr)
        pl_plan(M.seq, TP.B, TP.S, TP.ncu, TP.max_tiles, TP.split_long, TP.count, TP.plan, M.SP, reinterpret_cast<unsigned char*>(lds), PL_MODE_SPANS, M.epoch,
                (int)blockIdx.x - ((int)gridDim.x - parts), parts);
    __syncthreads();
}

template <int D, int HS>
__global__ __launch_bounds__(SO_NT) void enc_tail_k(const float* __restrict__ g, const int32_t* __restrict__ keys, int nreg, int64_t stride,
                                                    const int32_t* __restrict__ n_dev, int n_mul, int64_t n_host, int64_t R, int rpw,
                                                    int64_t padding_idx, float scale, float* __restrict__ dW, SoAdam AD, TailJobs J, TailPrep TP) {
    extern __shared__ __align__(16) float lds[];
    re_kernarg_warm<re_kernarg_bytes(&enc_tail_k<D, HS>)>();
#ifdef TAIL_PROFILE
    if (threadIdx.x < TAIL_MARKS && blockIdx.x < TAIL_MARK_WGS) g_tail_marks[blockIdx.x * TAIL_MARKS + threadIdx.x] = 0ull;
    __syncthreads();
#endif
    TAIL_MARK(0, TAIL_NOW());
    tail_spans(TP, lds);
    // (requested here, first needed by the jobs; the dependence of the key count on it -- tile counts are never negative -- keeps the compiler
    // from sinking the load to its first use, where it would be a round trip of its own)
    const int n_tiles = enc_plan_view(J.plan, J.B, J.S).hdr[1];
    n_host += n_tiles < 0 ? 1 : 0;
    n_mul += n_tiles < 0 ? 1 : 0;
    TailEarly early{J.ticket, (int)gridDim.x};
    so_body<D, HS>(g, keys, nreg, stride, n_dev, n_mul, n_host, R, rpw, padding_idx, scale, dW, AD, lds, early);
    TAIL_MARK(1, TAIL_NOW());
    tail_jobs<D>(J, TP, lds, n_tiles, early.has);
    TAIL_MARK(15, TAIL_NOW());
}

// the same behind the row-sparse Adam of a LARGE table (adam_rows_owner.h; config 5: D = 128, HS = 2)
template <int D, int HS>
__global__ __launch_bounds__(SA_NT) void enc_tail_sparse_k(SaParams P, TailJobs J, TailPrep TP) {
    extern __shared__ __align__(16) float lds[];
    re_kernarg_warm<re_kernarg_bytes(&enc_tail_sparse_k<D, HS>)>();
#ifdef TAIL_PROFILE
    if (threadIdx.x < TAIL_MARKS && blockIdx.x < TAIL_MARK_WGS) g_tail_marks[blockIdx.x * TAIL_MARKS + threadIdx.x] = 0ull;
    __syncthreads();
#endif
    TAIL_MARK(0, TAIL_NOW());
    tail_spans(TP, lds);
    // the tape's hand-over error word (a tile waited for a partner's rows in vain: this step's gradients are wrong): the table's rows stay as
    // they are -- read here, on the device, every step; the epoch's check_handover() reports it
    const unsigned gated = reinterpret_cast<const unsigned*>(J.tape + J.T.off_FLAGS)[(J.NR / 16) * EP_FLAG_WORDS];
    if (!gated) sa_body<1, HS, int32_t>(P, reinterpret_cast<unsigned char*>(lds));
    TAIL_MARK(1, TAIL_NOW());
    tail_jobs<D>(J, TP, lds, enc_plan_view(J.plan, J.B, J.S).hdr[1]);
    TAIL_MARK(15, TAIL_NOW());
}

size_t enc_wgrad_part_floats(int64_t D, int64_t L);
size_t enc_wgrad_ppart_floats(int64_t B, int64_t D);
extern "C" size_t re_sasrec_encoder_bwd_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t L);
int enc_grad_reduce_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* plan, const float* slab, int nwg, const float* part,
                           const float* ppart, float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b,
                           hipStream_t s, int by_tile, const re_adam_fuse* adam, unsigned* ticket, const unsigned* gate, int nsplit);

static int tail_prep(TailPrep& TP, const re_next_prep* next) {
    TP = TailPrep{};
    if (!next) return RE_OK;
    if (!next->mail || !next->plan || next->B <= 0 || next->S <= 0 || next->S > 64) return RE_EINVAL;
    if (next->plan_bytes < enc_plan_bytes(next->B, next->S)) return RE_EWORKSPACE;
    if (next->B > PL_LDS_B) return RE_EUNSUPPORTED;          // (the plan job keeps spans and placements in LDS)
    const bool elementwise = next->seq_out || next->valid || next->rows_all || next->pos_out;
    TP.mail = (const PlMail*)next->mail;
    TP.B = (int)next->B; TP.S = (int)next->S; TP.ncu = next->ncu < 1 ? 1 : next->ncu; TP.max_tiles = next->max_tiles; TP.split_long = next->split_long;
    TP.n_ew = elementwise ? (int)re_grid(next->B * next->S, PL_NT, 256) : 0;
    TP.span_parts = (int)(next->B <= 512 ? 1 : next->B >= 16 * 512 ? 16 : (next->B + 511) / 512);
    TP.seq_out = (int64_t*)next->seq_out; TP.pos_out = (int64_t*)next->pos_out; TP.neg_out = (int64_t*)next->neg_out;
    TP.valid = (uint8_t*)next->valid; TP.count = (int*)next->count; TP.rows_all = (int64_t*)next->rows_all; TP.plan = (int*)next->plan;
    return RE_OK;
}

// the weight-gradient side of both entry points: checks, and the workspace as re_sasrec_encoder_step_part lays it out
struct TailSide {
    TailJobs J;
    float *slab, *wpart, *ppart;
    int wgrid;
    const unsigned* gate;   // the tape's hand-over error word (csrc/enc_tile_body.inc: tl_flag_wait): set = this step's gradients are not to be applied
};
static int tail_side(TailSide& T, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L, const void* plan, int32_t ncu, const void* tape,
                     size_t tape_bytes, const float* dx0, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b, void* ws,
                     size_t ws_bytes, uint32_t* ticket) {
    if (!seq || !plan || !tape || !dx0 || !block_grads || !g_last_w || !g_last_b || !ws || !ticket || B < 0) return RE_EINVAL;
    if ((D != 64 && D != 128) || S < 1 || S > 64 || L < 1 || L > SE_MAX_BLOCKS) return RE_EUNSUPPORTED;
    if (tape_bytes < (size_t)enc_tape_layout(B, S, D, L).total * sizeof(float) || ws_bytes < re_sasrec_encoder_bwd_workspace_bytes(B, S, D, L))
        return RE_EWORKSPACE;
    for (int64_t i = 0; i < 12 * L; ++i)
        if (!block_grads[i]) return RE_EINVAL;
    if (ncu < 1) ncu = 256;
    const int64_t mt = enc_plan_max_tiles(B, S);
    T.wgrid = (int)(mt < ncu ? mt : ncu);
    const EncBwdWs Wk = enc_bwd_ws(ws, B, S, D, L);
    T.slab = Wk.slab; T.wpart = Wk.wpart; T.ppart = Wk.ppart;
    float* gtape = Wk.gtape;
    T.J = TailJobs{(const float*)tape, enc_tape_layout(B, S, D, L), gtape, 16 * mt, plan, (int)B, (int)S, (int)L, T.wpart, seq, dx0,
                   dPtab ? T.ppart : nullptr, wg_nsplit_tail((int)D, B), ticket};
    T.gate = reinterpret_cast<const unsigned*>((const float*)tape + T.J.T.off_FLAGS) + mt * EP_FLAG_WORDS;
    return RE_OK;
}

extern "C" int re_sasrec_step_tail(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev, int32_t n_mul,
                                   int64_t R, int64_t padding_idx, float* dW, const re_adam_fuse* table_adam, const int64_t* seq, int64_t B,
                                   int64_t S, int64_t D, int64_t L, const void* plan, int32_t ncu, const void* tape, size_t tape_bytes,
                                   const float* dx0, float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b,
                                   void* ws, size_t ws_bytes, const re_adam_fuse* enc_adam, uint32_t* ticket, const re_next_prep* next,
                                   re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    TailPrep TP;
    const int rcp = tail_prep(TP, next);
    if (rcp != RE_OK) return rcp;
    if (!g || !keys || !n_dev || (!dW && !table_adam) || R <= 0 || n_regions < 1 || n_regions > 4 || region_stride < 0 || n_mul < 1) return RE_EINVAL;
    if (D != 64) return RE_EUNSUPPORTED;
    if (table_adam && (!table_adam->param || !table_adam->m || !table_adam->v || !table_adam->hyper)) return RE_EINVAL;
    SoAdam AD{};
    if (table_adam)
        AD = SoAdam{table_adam->param, table_adam->m, table_adam->v, table_adam->hyper, (float)table_adam->beta1, (float)table_adam->beta2,
                    (float)(1.0 - table_adam->beta1), (float)(1.0 - table_adam->beta2), (float)table_adam->eps, (float)table_adam->weight_decay};
    if ((region_stride & 3) || ((reinterpret_cast<uintptr_t>(keys) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(dW) |
                                 reinterpret_cast<uintptr_t>(AD.W) | reinterpret_cast<uintptr_t>(AD.m) | reinterpret_cast<uintptr_t>(AD.v)) & 15u))
        return RE_EUNSUPPORTED;
    if ((int64_t)n_regions * region_stride >= (1ll << 25)) return RE_EUNSUPPORTED;
    TailSide T;
    const int rc = tail_side(T, seq, B, S, D, L, plan, ncu, tape, tape_bytes, dx0, dPtab, block_grads, g_last_w, g_last_b, ws, ws_bytes, ticket);
    if (rc != RE_OK) return rc;
    AD.gate = T.gate;      // a hand-over that timed out in this step's tile kernel: gradients are written, neither optimizer moves (read ON THE DEVICE, every step)
    constexpr int HS = 2;
    const int rpw = 96;                            // (scatter.hip: scatter_small_launch)
    int64_t nwg = HS;
    while (nwg * rpw < R * HS) nwg *= 2;
    if (nwg > 4096) return RE_EUNSUPPORTED;
    const size_t lds_scatter = (size_t)SO_NG * rpw * (D / HS) * sizeof(float), lds_jobs = (size_t)2 * wg_job_lds_floats<64>() * sizeof(float);
    size_t ldsb = lds_scatter > lds_jobs ? lds_scatter : lds_jobs;
    if (ldsb < (size_t)PL_LDS_BYTES) ldsb = PL_LDS_BYTES;
    hipStream_t s = (hipStream_t)stream;
    auto k = enc_tail_k<64, HS>;
    if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(SO_NT), ldsb, s, g, keys, (int)n_regions, region_stride, n_dev, (int)n_mul, (int64_t)0, R, rpw,
                       padding_idx, 1.0f, dW, AD, T.J, TP);
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    return enc_grad_reduce_launch(B, S, D, L, plan, T.slab, T.wgrid, T.wpart, T.ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, s, 1, enc_adam,
                                  ticket, T.gate, T.J.nsplit);
}

// The same behind re_sparse_adam_rows_small (int32 keys, hyper from device memory): the tail of a LARGE-table step (config 5), D = 64 or 128.
extern "C" int re_sasrec_step_tail_sparse(const float* g, const int32_t* keys, int32_t n_regions, int64_t region_stride, const int32_t* n_dev,
                                          int64_t n_mul, int64_t R, int64_t padding_idx, float* W, float* m, float* v, const float* hyper, double beta1,
                                          double beta2, double eps, double weight_decay, const int64_t* seq, int64_t B, int64_t S, int64_t D, int64_t L,
                                          const void* plan, int32_t ncu, const void* tape, size_t tape_bytes, const float* dx0, float emb_scale,
                                          float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b, void* ws, size_t ws_bytes,
                                          const re_adam_fuse* enc_adam, uint32_t* ticket, const re_next_prep* next, re_stream_t stream) {
    re_clear_error();
    if (B == 0) return RE_OK;
    TailPrep TP;
    const int rcp = tail_prep(TP, next);
    if (rcp != RE_OK) return rcp;
    if (!g || !keys || !n_dev || !W || !m || !v || !hyper || R <= 0 || n_regions < 1 || region_stride <= 0 || n_mul < 1) return RE_EINVAL;
    if (R >= 0xFFFFFFFEll || (int64_t)n_regions * region_stride >= 0xFFFFFFFFll) return RE_EUNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 7u) return RE_EINVAL;
    TailSide T;
    const int rc = tail_side(T, seq, B, S, D, L, plan, ncu, tape, tape_bytes, dx0, dPtab, block_grads, g_last_w, g_last_b, ws, ws_bytes, ticket);
    if (rc != RE_OK) return rc;
    SaParams P;
    P.g = g; P.keys = keys; P.n_regions = n_regions; P.region_stride = region_stride; P.n_dev = n_dev; P.n_mul = n_mul; P.n_host = 0;
    P.R = R; P.padding_idx = padding_idx; P.W = W; P.m = m; P.v = v;
    P.b1 = (float)beta1; P.b2 = (float)beta2; P.omb1 = (float)(1.0 - beta1); P.omb2 = (float)(1.0 - beta2);
    P.eps = (float)eps; P.wd = (float)weight_decay; P.hyper = hyper;
    P.step_size = 0.f; P.inv_sqrt_bc2 = 0.f;
    P.stride = D; P.coff = 0;
    hipStream_t s = (hipStream_t)stream;
    if (D == 128) {
        const size_t lds_jobs = (size_t)2 * wg_job_lds_floats<128>() * sizeof(float);
        const size_t ldsb = lds_jobs > (size_t)SA_LDS_BYTES(1) ? lds_jobs : (size_t)SA_LDS_BYTES(1);   // (> PL_LDS_BYTES)
        auto k = enc_tail_sparse_k<128, 2>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(SA_NWG), dim3(SA_NT), ldsb, s, P, T.J, TP);
    } else {
        const size_t lds_jobs = (size_t)2 * wg_job_lds_floats<64>() * sizeof(float);
        const size_t ldsb = lds_jobs > (size_t)SA_LDS_BYTES(1) ? lds_jobs : (size_t)SA_LDS_BYTES(1);
        auto k = enc_tail_sparse_k<64, 1>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(SA_NWG), dim3(SA_NT), ldsb, s, P, T.J, TP);
    }
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    return enc_grad_reduce_launch(B, S, D, L, plan, T.slab, T.wgrid, T.wpart, T.ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, s, 1, enc_adam,
                                  ticket, T.gate, T.J.nsplit);
}
