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
// 1: the reduction as the queue's last jobs (TailReduce).  Built, bit-identical (tests/test_gpu_sasrec.py run both ways), and NOT faster: the
// launch ends one reduction (a memory round trip deep) behind its SLOWEST matrix job either way, and with ~50 more workgroups alive the slowest
// matrix job itself ends later (ticket durations: median 26 k cycles both ways, max 37 k -> 49 k): 37 - 42 us against 25.5 + 6.9 for the two
// launches (scripts/tail_phases.py, profiles/r5_tail_reduce_in_queue.txt).  So: 0, the reduction is the launch behind the tail.
#ifndef TAIL_REDUCE_IN_QUEUE
#define TAIL_REDUCE_IN_QUEUE 0
#endif
#include "enc_grad_reduce.h"
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
    unsigned* ticket;    // FOUR counters, TK_STRIDE words (a cache line) apart, zero at launch: [0] the queue's ticket counter, [1] matrix jobs done,
                         // [2] position jobs done, [3] workgroups that left the queue.  With the reduction in the queue (TailReduce.on) the LAST
                         // workgroup to leave zeroes all four; without it enc_grad_reduce_k (the next launch) zeroes [0].  (A line each: with the four
                         // in one line the ~50 workgroups polling a done count kept that line busy and every ticket's atomic queued behind them --
                         // the launch took 68 us instead of 25.)
};

// The reduction (enc_grad_reduce.h) as the queue's last jobs: 1024-thread workgroups run FOUR of its 256-thread virtual blocks per ticket.  The
// matrix blocks wait until every matrix job's partial is in memory (word [1] of the ticket area; a ticket is only handed out when all tickets
// before it have been taken, so the jobs waited for are running on resident workgroups or done: no deadlock), the position blocks likewise
// (word [2]); the vector blocks read what the step's item / tile kernel left.  One launch, its dispatch and its ~7 us less per step; the sums
// and their order are the launch's own (bit for bit).
#define TK_STRIDE 32
struct TailReduce {
    EgReduce R;
    int on;
};

#if TAIL_REDUCE_IN_QUEUE
__shared__ TailReduce s_tr;   // the reduction's arguments, copied out of the argument segment once per workgroup (the call below takes a pointer)

__device__ __attribute__((noinline)) void tail_reduce_job(unsigned* ticket, const EgReduce* Rp, int t, int n_mat,
                                                                                                    int n_pos, int n_rmat, int n_rvec, float* lds) {
    const EgReduce& R = *Rp;
    const int tid = threadIdx.x, sub = tid >> 8, t256 = tid & 255;   // sub: the thread's 256-thread virtual block
    if (t < n_rmat || t >= n_rmat + n_rvec) {
        const bool pos = t >= n_rmat;
        const unsigned want = pos ? (unsigned)n_pos : (unsigned)n_mat;
        if (tid == 0) {
            const unsigned* w = ticket + (pos ? 2 : 1) * TK_STRIDE;
            while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
        if (!pos) { if (4 * t + sub < R.nmat_blocks) eg_reduce_mat<true>(R, 4 * t + sub, t256); }
        else { const int vb = 4 * (t - n_rmat - n_rvec) + sub; if (vb < R.npos_blocks) eg_reduce_pos<true>(R, vb, t256); }
    } else {
        float (*red)[64] = reinterpret_cast<float (*)[64]>(lds + sub * 256);
        const EgVec V = eg_reduce_vec_a(R, 4 * (t - n_rmat) + sub, t256, red);
        __syncthreads();
        eg_reduce_vec_b(R, V, t256, red);
    }
}
#endif

// The NEXT batch's preparation (enc_plan_body.h: it depends on the batch alone) as the first jobs of the ticket queue: one plan job, n_ew
// element-wise jobs -- in front of a step it is a 14 us launch (one workgroup's chain of barriers and round trips), here it rides in
// workgroups that are done with the table while others still work.  mail: a PlMail in device memory (the next batch's tensors or its sampling
// source), written by re_sasrec_step_stage[_sample] in front of this step (neither: no next batch); outputs: the OTHER captured copy's buffers.
struct TailPrep {
    const PlMail* mail;           // nullptr: this launch prepares nothing
    int B, S, ncu, max_tiles, split_long, n_ew;
    int64_t *seq_out, *pos_out, *neg_out;
    uint8_t* valid;
    int* count;
    int64_t* rows_all;
    int* plan;
};
static_assert(PL_NT == SO_NT && PL_NT == SA_NT, "the preparation jobs are written for the tail launches' workgroup size");

// ---- weight-gradient jobs.  D = 64: two per ticket: matrix jobs q = 2 t + half -> (block, matrix, split) = (q / 144, q / 24 % 6, q % 24); D = 128:
//      one per ticket, run by the whole workgroup: (t / 72, t / 12 % 6, t % 12).  Then the position-table jobs (144 strides of the (position,
//      chunk) list, two per ticket) -- both halves of a workgroup always run the same kind
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
#if TAIL_REDUCE_IN_QUEUE
#define TR_ON(TR) ((TR).on)
#define TAIL_TR_PARAM , TailReduce TR
#define TAIL_TR_LOCAL
#define TAIL_TR_ARG , TR
#else
#define TR_ON(TR) false
#define TAIL_TR_PARAM                     /* (no argument block for it: 620 bytes more of kernel arguments cost the launch ~2 us) */
#define TAIL_TR_LOCAL const TailReduce TR{};
#define TAIL_TR_ARG
#endif
template <int D>
__device__ __forceinline__ void tail_jobs(const TailJobs& J, const TailPrep& TP, const TailReduce& TR, float* lds, int n_tiles, bool has_early = false) {
    const int tid = threadIdx.x, half = tid >> 9, ht = tid & 511;
    float* jl = lds + half * wg_job_lds_floats<D>();
    constexpr int WG_NSPLIT = wg_nsplit(D);
    constexpr int PER_PLANE = WG_NSPLIT * EG_NMAT;      // matrix jobs of a block
    constexpr bool WHOLE = D == 128;                    // a matrix job is run by the whole workgroup (enc_wgrad_job.h: wg_nsplit), not two by its halves
    constexpr int POS_GROUPS = 144;                     // the position jobs are dealt to this many 512-thread groups (two per ticket)
    const int n_mat = WHOLE ? J.L * PER_PLANE : J.L * PER_PLANE / 2, n_pos = J.ppart ? POS_GROUPS / 2 : 0;
    const int n_prep = TP.mail ? 1 + TP.n_ew : 0;
    const int n_rmat = TR_ON(TR) ? (TR.R.nmat_blocks + 3) / 4 : 0, n_rvec = TR_ON(TR) ? (TR.R.nvec_blocks + 3) / 4 : 0, n_rpos = TR_ON(TR) ? (TR.R.npos_blocks + 3) / 4 : 0;
    int jn = 0;
    int done_kind = 0;   // (uniform) the job just finished: 1 a matrix job, 2 a position job -- counted once its stores have drained
    for (;;) {
        if (TR_ON(TR) && done_kind) {
            re_sync_full();                                       // every thread's partial stores (agent scope: written through) have completed ...
            if (tid == 0) atomicAdd(J.ticket + done_kind * TK_STRIDE, 1u);   // ... before the job counts as done
            done_kind = 0;
        }
        if (jn == 0) TJ_STAMP(3);
        __syncthreads();   // (the launch's first part / the previous job's stages are done with the LDS)
        if (jn > 0 && jn <= 3) TAIL_MARK(1 + 2 * jn, TAIL_NOW());
        if (jn == 0) TJ_STAMP(0);
        if (tid == 0 && !(has_early && jn == 0)) s_job = (int)atomicAdd(J.ticket, 1u);
        __syncthreads();
        int t = s_job;
        if (jn == 0) TJ_STAMP(1);
        if (t >= n_prep + n_rvec + n_mat + n_pos + n_rmat + n_rpos) break;
        if (jn < 3) TAIL_MARK(2 + 2 * jn, (unsigned long long)(t + 1));
        ++jn;
        if (t < n_prep) {
            const PlMail M = *TP.mail;
            if (M.seq || M.SP.ptr) {                           // (uniform)
                if (t == 0) pl_plan(M.seq, TP.B, TP.S, TP.ncu, TP.max_tiles, TP.split_long, TP.count, TP.plan, M.SP, reinterpret_cast<unsigned char*>(lds), PL_MODE_REST, M.epoch);
                else pl_elementwise(t - 1, TP.n_ew, M.seq, M.pos, M.neg, TP.B, TP.S, TP.seq_out, TP.pos_out, TP.neg_out, TP.valid, TP.rows_all, M.SP);
            }
            continue;
        }
        t -= n_prep;
        // (the vector gradients' reduction depends on the item / tile kernel alone and is a chain of memory round trips: its tickets come FIRST;
        //  the matrix and position reductions wait for this launch's jobs: LAST)
#if TAIL_REDUCE_IN_QUEUE
        if (t < n_rvec) {
            tail_reduce_job(J.ticket, &s_tr.R, n_rmat + t, n_mat, n_pos, n_rmat, n_rvec, lds);
            continue;
        }
#endif
        t -= n_rvec;
        if (t >= n_mat + n_pos) {
            // ---- the reduction's blocks, four per ticket: a CALL, not inlined -- inlined, its registers (24 partials in flight per thread, the
            //      argument block's scalars) raised the pressure in the matrix jobs' loops of this 128-register kernel: scratch 28 -> 116 bytes a
            //      lane, and the launch took 57 us instead of 25 + 7
            TAIL_MARK(24, TAIL_NOW());
#if TAIL_REDUCE_IN_QUEUE
            {
                const int r = t - n_mat - n_pos;                 // [0, n_rmat): matrices; then the position table's
                tail_reduce_job(J.ticket, &s_tr.R, r < n_rmat ? r : r + n_rvec, n_mat, n_pos, n_rmat, n_rvec, lds);
            }
#endif
            TAIL_MARK(25, TAIL_NOW());
            TAIL_MARK(26, (unsigned long long)(t - n_mat - n_pos + 1));
            continue;
        }
        if (t < n_mat) {
            if (jn == 1) TJ_STAMP(2);
            done_kind = 1;
            if constexpr (WHOLE) {
                wg_matrix_job<D, SO_NT>(tid, lds, t / PER_PLANE, (t / WG_NSPLIT) % EG_NMAT, t % WG_NSPLIT, J.tape, J.T, J.gtape, J.NR, n_tiles, J.part);
            } else {
                const int q = 2 * t + half;
                wg_matrix_job<D>(ht, jl, q / PER_PLANE, (q / WG_NSPLIT) % EG_NMAT, q % WG_NSPLIT, J.tape, J.T, J.gtape, J.NR, n_tiles, J.part);
            }
        } else {
            wg_pos_job<D>(ht, jl, 2 * (t - n_mat) + half, POS_GROUPS, J.B, J.S, J.seq, J.contrib, J.ppart);
            done_kind = 2;
        }
    }
    if (TR_ON(TR) && tid == 0) {
        // every workgroup of the grid passes here once; the last one leaves the four words zero for the next step
        if (atomicAdd(J.ticket + 3 * TK_STRIDE, 1u) == gridDim.x - 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) __hip_atomic_store(J.ticket + k * TK_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

#if TAIL_REDUCE_IN_QUEUE
// (a barrier follows before anybody reads it: both kernels' first parts end in one)
__device__ __forceinline__ void tail_stash_reduce(const TailReduce& TR) {
    if (!TR_ON(TR)) return;
    constexpr int NW = (int)(sizeof(TailReduce) / 4);
    const unsigned* src = reinterpret_cast<const unsigned*>(&TR);
    unsigned* dst = reinterpret_cast<unsigned*>(&s_tr);
    for (int i = threadIdx.x; i < NW; i += blockDim.x) dst[i] = src[i];
}
#else
__device__ __forceinline__ void tail_stash_reduce(const TailReduce&) {}
#endif

// The next batch's spans (the plan's phase 1: the only part that reads the batch) by the launch's LAST workgroup, from the launch's start:
// the plan job of the queue (PL_MODE_REST) then starts ~23 k cycles in with the spans already there, and ends with the weight-gradient tickets
// instead of ~17 k cycles behind them (scripts/tail_phases.py).  The last workgroup owns cold rows (popular items have small ids) and, coming
// late to the queue, takes no ticket: its extra work is hidden.
__device__ __forceinline__ void tail_spans(const TailPrep& TP, float* lds) {
    if (!TP.mail || blockIdx.x != gridDim.x - 1) return;
    const PlMail M = *TP.mail;
    if (M.seq || M.SP.ptr)
        pl_plan(M.seq, TP.B, TP.S, TP.ncu, TP.max_tiles, TP.split_long, TP.count, TP.plan, M.SP, reinterpret_cast<unsigned char*>(lds), PL_MODE_SPANS, M.epoch);
    __syncthreads();
}

template <int D, int HS>
__global__ __launch_bounds__(SO_NT) void enc_tail_k(const float* __restrict__ g, const int32_t* __restrict__ keys, int nreg, int64_t stride,
                                                    const int32_t* __restrict__ n_dev, int n_mul, int64_t n_host, int64_t R, int rpw,
                                                    int64_t padding_idx, float scale, float* __restrict__ dW, SoAdam AD, TailJobs J, TailPrep TP TAIL_TR_PARAM) {
    TAIL_TR_LOCAL
    extern __shared__ __align__(16) float lds[];
    re_kernarg_warm<re_kernarg_bytes(&enc_tail_k<D, HS>)>();
    tail_stash_reduce(TR);
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
    tail_jobs<D>(J, TP, TR, lds, n_tiles, early.has);
    TAIL_MARK(15, TAIL_NOW());
}

// the same behind the row-sparse Adam of a LARGE table (adam_rows_owner.h; config 5: D = 128, HS = 2)
template <int D, int HS>
__global__ __launch_bounds__(SA_NT) void enc_tail_sparse_k(SaParams P, TailJobs J, TailPrep TP TAIL_TR_PARAM) {
    TAIL_TR_LOCAL
    extern __shared__ __align__(16) float lds[];
    re_kernarg_warm<re_kernarg_bytes(&enc_tail_sparse_k<D, HS>)>();
    tail_stash_reduce(TR);
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
    tail_jobs<D>(J, TP, TR, lds, enc_plan_view(J.plan, J.B, J.S).hdr[1]);
    TAIL_MARK(15, TAIL_NOW());
}

size_t enc_wgrad_part_floats(int64_t D, int64_t L);
size_t enc_wgrad_ppart_floats(int64_t B, int64_t D);
extern "C" size_t re_sasrec_encoder_bwd_workspace_bytes(int64_t B, int64_t S, int64_t D, int64_t L);
int enc_grad_reduce_launch(int64_t B, int64_t S, int64_t D, int64_t L, const void* plan, const float* slab, int nwg, const float* part,
                           const float* ppart, float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b,
                           hipStream_t s, int by_tile, const re_adam_fuse* adam, unsigned* ticket, const unsigned* gate);

int enc_grad_reduce_args(EgReduce& R, int64_t B, int64_t S, int64_t D, int64_t L, const void* plan, const float* slab, int nwg, const float* part,
                         const float* ppart, float emb_scale, float* dPtab, float* const* block_grads, float* g_last_w, float* g_last_b, int by_tile,
                         const re_adam_fuse* adam, const unsigned* gate);

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
    T.slab = (float*)ws;
    T.wpart = T.slab + (size_t)enc_slab_rows(B, S) * L * EG_NVEC * D;
    T.ppart = T.wpart + enc_wgrad_part_floats(D, L);
    float* gtape = T.ppart + enc_wgrad_ppart_floats(B, D);
    T.J = TailJobs{(const float*)tape, enc_tape_layout(B, S, D, L), gtape, 16 * mt, plan, (int)B, (int)S, (int)L, T.wpart, seq, dx0,
                   dPtab ? T.ppart : nullptr, ticket};
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
    TailReduce TR{};
    TR.on = TAIL_REDUCE_IN_QUEUE;
    if (TR.on) {
        const int rcr = enc_grad_reduce_args(TR.R, B, S, D, L, plan, T.slab, T.wgrid, T.wpart, T.ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, 1,
                                             enc_adam, T.gate);
        if (rcr != RE_OK) return rcr;
    }
    hipLaunchKernelGGL(k, dim3((unsigned)nwg), dim3(SO_NT), ldsb, s, g, keys, (int)n_regions, region_stride, n_dev, (int)n_mul, (int64_t)0, R, rpw,
                       padding_idx, 1.0f, dW, AD, T.J, TP TAIL_TR_ARG);
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    if (TR.on) return RE_OK;
    return enc_grad_reduce_launch(B, S, D, L, plan, T.slab, T.wgrid, T.wpart, T.ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, s, 1, enc_adam,
                                  ticket, T.gate);
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
    TailReduce TR{};
    TR.on = TAIL_REDUCE_IN_QUEUE;
    if (TR.on) {
        const int rcr = enc_grad_reduce_args(TR.R, B, S, D, L, plan, T.slab, T.wgrid, T.wpart, T.ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, 1,
                                             enc_adam, T.gate);
        if (rcr != RE_OK) return rcr;
    }
    if (D == 128) {
        const size_t lds_jobs = (size_t)2 * wg_job_lds_floats<128>() * sizeof(float);
        const size_t ldsb = lds_jobs > (size_t)SA_LDS_BYTES(1) ? lds_jobs : (size_t)SA_LDS_BYTES(1);   // (> PL_LDS_BYTES)
        auto k = enc_tail_sparse_k<128, 2>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(SA_NWG), dim3(SA_NT), ldsb, s, P, T.J, TP TAIL_TR_ARG);
    } else {
        const size_t lds_jobs = (size_t)2 * wg_job_lds_floats<64>() * sizeof(float);
        const size_t ldsb = lds_jobs > (size_t)SA_LDS_BYTES(1) ? lds_jobs : (size_t)SA_LDS_BYTES(1);
        auto k = enc_tail_sparse_k<64, 1>;
        if (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb) != hipSuccess) return RE_ELAUNCH;
        hipLaunchKernelGGL(k, dim3(SA_NWG), dim3(SA_NT), ldsb, s, P, T.J, TP TAIL_TR_ARG);
    }
    if (hipGetLastError() != hipSuccess) return RE_ELAUNCH;
    if (TR.on) return RE_OK;
    return enc_grad_reduce_launch(B, S, D, L, plan, T.slab, T.wgrid, T.wpart, T.ppart, emb_scale, dPtab, block_grads, g_last_w, g_last_b, s, 1, enc_adam,
                                  ticket, T.gate);
}
