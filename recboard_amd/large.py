"""SASRec on an item table too large for dense gradients / dense Adam (BASELINE config 5: 100 M items x D = 128).

Same model as `recboard_amd.sasrec.SASRecEngine` (reference `SASRec/main.py:63-228`), different memory plan:

  * the item table E [N+1, D] and its Adam moments are plain tensors outside the parameter arena (51 GB + 2 x 51 GB at
    N = 1e8, D = 128: fits one MI355X's 288 GB); no [N+1, D] gradient ever exists;
  * a step produces the 3*B*S item-gradient CONTRIBUTION ROWS (d x0 through the embedding backward, and the criterion's
    positive / negative rows) and hands them, with their destination rows, to `re_sparse_adam_rows`: deterministic segmented
    sums + one Adam update per distinct row (torch.optim.SparseAdam's rule + coupled weight decay on the touched rows) --
    a dense Adam over the table would move 28 bytes per element, 358 GB per step;
  * everything else (position table, blocks, lastLN) stays in one dense arena with one fused Adam launch.

Encoder: the HIP kernels (D = 64 and 128; at D = 128 a work item holds 32 rows in LDS and longer sequences are taken in chained
parts, csrc/enc_fwd.hip) with the embedding front end / backward fused in; any other shape raises (RE_EUNSUPPORTED) -- the engine has no
torch encoder (a torch restatement of the block stack is a comparator in tests/aten_sasrec.py).
Multi-GPU: `recboard_amd.sharded.ShardedTable.lookup / backward_sparse_adam` is the same step with the table row-sharded
(one all-to-all per direction); this class is the single-GPU form.
"""
import math
from collections import OrderedDict

import torch

from . import ops
from .capture import recording
from .sasrec import ParamArena, SASRecEngine, param_shapes


def counter_normal_rows(rows, D, seed, std, device):
    """Table rows as a pure function of (seed, global row id, column): N(0, std^2) from a 32-bit counter hash, so that a
    table initialised shard by shard holds the same values for every GPU count.  `rows` int64 [r]."""
    M = 0xFFFFFFFF
    idx = (rows.to(torch.int64).unsqueeze(1) * D + torch.arange(D, device=device, dtype=torch.int64)) & 0xFFFFFFFFFFFF

    def mix(h):   # murmur3's finaliser on the low 32 bits (int64 arithmetic, masked)
        h = h & M
        h = h ^ (h >> 16)
        h = (h * 0x85EBCA6B) & M
        h = h ^ (h >> 13)
        h = (h * 0xC2B2AE35) & M
        return h ^ (h >> 16)

    lo = mix((idx & M) * 0x9E3779B1 + (idx >> 32) * 0x7FEB352D + (seed & M))
    hi = mix(lo * 0x85EBCA77 + 0x165667B1 + (seed & M))
    u = ((lo << 21) ^ hi).to(torch.float64) * (1.0 / float(1 << 53)) + (0.5 / float(1 << 53))   # (0, 1), 53 bits
    return (math.sqrt(2.0) * std * torch.erfinv(2.0 * u - 1.0)).to(torch.float32)


class SASRecLargeTableEngine(SASRecEngine):
    def __init__(self, num_items, maxlen=50, embedding_dim=128, num_blocks=2, dropout_rate=0.0, loss="BCE", lr=1e-3,
                 weight_decay=0.0, betas=(0.9, 0.999), device="cuda", seed=1, table_std=0.02, table_init="torch", encoder=None):
        assert loss in ("BCE", "BPR")
        # the encoder kernels cover D = 64 and 128 (BASELINE config 5 is D = 128)
        if (encoder or "fused") != "fused" or embedding_dim not in (64, 128) or maxlen > 64 or num_blocks > 4:
            raise NotImplementedError(f"recengine: unsupported shape (RE_EUNSUPPORTED): the encoder kernels cover D = 64 or 128, maxlen <= 64, "
                                      f"blocks <= 4; got D = {embedding_dim}, maxlen = {maxlen}, blocks = {num_blocks}, encoder = {encoder!r}")
        self.encoder = "fused"
        self.compact_rows = True     # fused encoder: the step on the batch plan's compact rows (criterion in the forward kernel)
        self.split_long = True       # sequences of 3 - 4 tiles as two work items in two workgroups (at D = 128 a whole long item is two
                                     # SEQUENTIAL parts in one workgroup: the launch lasts twice a part)
        self.fused_item_kernel = True
        self._bufs = {}
        self.N, self.S, self.D, self.L = num_items, maxlen, embedding_dim, num_blocks
        self.p_drop, self.loss_kind = dropout_rate, loss
        self.lr, self.wd, self.betas = lr, weight_decay, betas
        self.device = torch.device(device)
        shapes = param_shapes(num_items, maxlen, embedding_dim, num_blocks)
        del shapes["Item.embeddings.weight"]                       # lives outside the arena
        self.arena = ParamArena(shapes, self.device)
        self.training = True
        self.seed = seed
        self.params = OrderedDict()
        for k in self.arena.shapes:
            self.params[k] = self.arena.view(self.arena.data, k).requires_grad_(True)
        self.table_std, self.table_init = table_std, table_init
        self._alloc_table(seed)
        self.reset_parameters(seed)

    def _alloc_table(self, seed):
        # item table + moments; row 0 = padding.  Filled in place, chunk-wise (no second table-sized temporary).
        R, D = self.N + 1, self.D
        self.E = torch.empty((R, D), dtype=torch.float32, device=self.device)
        g = torch.Generator(device=self.device).manual_seed(seed)
        step_rows = max(1, (1 << 24) // D)
        for r0 in range(0, R, step_rows):
            if self.table_init == "counter":   # values depend on (seed, row, column) only: identical for any sharding
                rows = torch.arange(r0, min(R, r0 + step_rows), device=self.device)
                self.E[r0:r0 + step_rows] = counter_normal_rows(rows, D, seed, self.table_std, self.device)
            else:
                self.E[r0:r0 + step_rows].normal_(0.0, self.table_std, generator=g)
        self.E[0].zero_()
        self.Em = torch.zeros_like(self.E)
        self.Ev = torch.zeros_like(self.E)

    def state_dict(self):
        sd = OrderedDict((k, p.detach().clone()) for k, p in self.params.items())
        sd["Item.embeddings.weight"] = self.E
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, p in self.params.items():
                p.copy_(torch.as_tensor(sd[k]).to(self.device).view(p.shape))
            self.E.copy_(torch.as_tensor(sd["Item.embeddings.weight"]).to(self.device))

    def reset_ranking_buffers(self):
        """Coach.evaluate calls this before a split's batches: split the item table once for the split's `recommend_topk` calls."""
        self._score_prep = (ops.score_prepare(self.E[1:]), self.arena.step)

    def optimizer_state(self):
        """Adam state incl. the table's moments (Coach.save_checkpoint; the arena alone would lose them on resume)."""
        return {"m": self.arena.m.clone(), "v": self.arena.v.clone(), "step": self.arena.step, "Em": self.Em, "Ev": self.Ev}

    def load_optimizer_state(self, st):
        self.arena.m.copy_(st["m"]); self.arena.v.copy_(st["v"]); self.arena.step = int(st["step"])
        if "Em" in st:
            self.Em.copy_(st["Em"]); self.Ev.copy_(st["Ev"])

    # ---- forward pieces
    def encode(self, seq):
        """-> (userEmbds [B,S,D], itemEmbds = E[1:]).  SASRec/main.py:178-193 (inference / evaluation)."""
        with torch.no_grad():
            P = self.params
            p = self.p_drop if self.training else 0.0
            u, _ = ops.sasrec_embed_encoder_fwd(self.E, P["Position.weight"].detach(), seq, float(self.D ** 0.5), self._block_tensors(),
                                                P["lastLN.weight"].detach(), P["lastLN.bias"].detach(), self.L, p, self._step_seed(),
                                                plan=ops.sasrec_plan(seq, self.D))
            return u, self.E[1:]

    def _grads(self, seq, pos, neg, aux, sd, seed_dev=None, table=None, adam_hyper=None, next_prep=None):
        """Forward + backward: encoder gradients into the arena, the item-gradient contribution rows C with their destination rows
        (0 = none).  -> (loss, C, rows).
        `table` (default: the item table) is what seq / pos / neg index: the sharded engine passes its batch-local table.
        adam_hyper (captured single-GPU step): BOTH optimizers too, in two branches -- the weight gradients and their reduction (which
        applies the encoder's dense Adam, re_adam_fuse) on a side stream beside the table's row-sparse Adam on this one; both depend on
        the item kernel alone.  -> (loss, C, rows, True) then."""
        A, D = self.arena, self.D
        B, S = seq.shape
        n = B * S
        valid, count = aux.valid, aux.count
        E = self.E if table is None else table
        p = self.p_drop if self.training else 0.0
        Ppos = self.params["Position.weight"]
        kind = ops.LOSS_BCE if self.loss_kind == "BCE" else ops.LOSS_BPR
        if self.encoder == "fused" and self.compact_rows:
            # SASRecEngine's compact-row step minus the dense table gradient: forward + criterion + backward of every work item in one
            # launch; what comes back are the 3 x NR contribution rows with their destination keys (0 = none) for the row-sparse Adam
            # (keys are rows of `table` when one is given: the sharded engine's batch-local table)
            W = self._buffers(B, S)
            G = A.views(A.grad)
            lw, lb = self.params["lastLN.weight"].detach(), self.params["lastLN.bias"].detach()
            if table is not None:                               # (the sharded step reads every key entry: rows beyond this batch's plan must read
                W["keys"].zero_()                               #  "no contribution"; the unsharded update stops at the plan's live length)
            if adam_hyper is not None and table is None and self.fused_item_kernel and getattr(self, "fuse_tail", True):
                # ONE queue: item kernels, then one launch in which the row-sparse Adam's workgroups go on with the weight-gradient jobs, then
                # the reduction (+ the encoder's dense Adam) -- csrc/enc_tail.hip (as two graph branches: fork and join cost ~20 us of a step)
                loss = ops.sasrec_encoder_step(E, Ppos.detach(), seq, pos, neg, float(D ** 0.5), self._block_tensors(), lw, lb, self.L, p, sd, aux.plan, kind,
                                               count, W["u"], W["tape"], W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"], W["contrib"][:n].view(B, S, D),
                                               G["Position.weight"], self._block_tensors(A.grad), G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"],
                                               e_off=1, seed_dev=seed_dev, part=3 + (8 if getattr(aux, "weights_ready", False) else 0))
                fz = ops.adam_fuse(A.grad, A.data, A.m, A.v, adam_hyper, self.betas[0], self.betas[1], 1e-8, self.wd)
                self._adam_keep = fz
                if not hasattr(self, "_ticket"):
                    self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
                ops.sasrec_step_tail_sparse(W["g_rows"].view(-1, D), W["keys"], self.E, self.Em, self.Ev, adam_hyper, self.betas[0], self.betas[1], 1e-8,
                                            self.wd, aux.plan.view(torch.int32)[1:2], 16, seq, self.L, aux.plan, W["tape"], W["contrib"][:n].view(B, S, D),
                                            float(D ** 0.5), G["Position.weight"], self._block_tensors(A.grad), G["lastLN.weight"], G["lastLN.bias"],
                                            W["ws_bwd"], self._ticket, enc_adam=fz, next=next_prep)
                return loss, W["g_rows"].view(-1, D), W["keys"], True
            if adam_hyper is not None and table is None and self.fused_item_kernel and getattr(self, "fork_wgrad", True):
                args = (E, Ppos.detach(), seq, pos, neg, float(D ** 0.5), self._block_tensors(), lw, lb, self.L, p, sd, aux.plan, kind, count, W["u"],
                        W["tape"], W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"], W["contrib"][:n].view(B, S, D), G["Position.weight"],
                        self._block_tensors(A.grad), G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"])
                main = torch.cuda.current_stream()
                if not hasattr(self, "_side"):
                    self._side = torch.cuda.Stream()
                    self._tail = torch.zeros(4, dtype=torch.int32, device=self.device)
                loss = ops.sasrec_encoder_step(*args, e_off=1, seed_dev=seed_dev, part=3 + (8 if getattr(aux, "weights_ready", False) else 0))
                self._side.wait_stream(main)
                fz = ops.adam_fuse(A.grad, A.data, A.m, A.v, adam_hyper, self.betas[0], self.betas[1], 1e-8, self.wd)
                self._adam_keep = fz
                with torch.cuda.stream(self._side):
                    ops.sasrec_encoder_step(*args, e_off=1, seed_dev=seed_dev, part=4, loss=loss, adam=fz)
                self._table_adam(W["g_rows"].view(-1, D), W["keys"], aux, hyper=adam_hyper)
                main.wait_stream(self._side)
                ops.step_state(self._tail, 0, 1, 1e-3)          # (a graph that ends in a join of two branches replays slower: one trivial node behind it)
                return loss, W["g_rows"].view(-1, D), W["keys"], True
            loss = ops.sasrec_encoder_step(E, Ppos.detach(), seq, pos, neg, float(D ** 0.5), self._block_tensors(), lw, lb, self.L, p, sd,
                                           aux.plan, kind, count, W["u"], W["tape"], W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"],
                                           W["contrib"][:n].view(B, S, D), G["Position.weight"], self._block_tensors(A.grad),
                                           G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"], e_off=1, seed_dev=seed_dev,
                                           part=8 if getattr(aux, "weights_ready", False) else 0)
            return loss, W["g_rows"].view(-1, D), W["keys"]          # keys: int32 [3, NR], the plan's first rows of every region live
        # the same launches as SASRecEngine's fused step, minus the dense scatter-add: fused embedding + encoder forward (tape),
        # criterion forward + backward, encoder backward with the embedding backward fused in (contribution rows, position gradient)
        W = self._buffers(B, S)
        G = A.views(A.grad)
        lw, lb = self.params["lastLN.weight"].detach(), self.params["lastLN.bias"].detach()
        bt = self._block_tensors()
        C = W["contrib"]
        ops.sasrec_embed_encoder_fwd(E, Ppos.detach(), seq, float(D ** 0.5), bt, lw, lb, self.L, p, sd, need_tape=True, out=W["u"],
                                     tape=W["tape"], plan=aux.plan, seed_dev=seed_dev)
        loss, _, _, _ = ops.pair_loss_fwd_bwd(W["u"].view(n, D), E, pos.reshape(-1), neg.reshape(-1), valid, kind, count, e_off=1,
                                              out=(W["dU"], C[n:2 * n], C[2 * n:]))
        ops.sasrec_encoder_embed_bwd(W["dU"].view(B, S, D), seq, float(D ** 0.5), bt, lw, lb, self.L, p, sd, W["tape"],
                                     self._block_tensors(A.grad), G["lastLN.weight"], G["lastLN.bias"], G["Position.weight"],
                                     out=C[:n].view(B, S, D), ws=W["ws_bwd"], plan=aux.plan, seed_dev=seed_dev)
        return loss, C, aux.rows_all

    def _table_adam(self, C, rows, aux, step=0, hyper=None):
        """Row-sparse Adam of the item table.  Compact-row step (rows: the int32 [3, NR] keys of the plan's rows): ONE launch, no sort
        (re_sparse_adam_rows_small; the live length of every region comes from the plan on the device); otherwise the sorted general form."""
        b1, b2 = self.betas
        if rows.dtype == torch.int32 and self.D in (64, 128):
            ops.sparse_adam_rows_small(C, rows, self.E, self.Em, self.Ev, step=step, lr=self.lr, beta1=b1, beta2=b2, eps=1e-8, weight_decay=self.wd,
                                       padding_idx=0, hyper=hyper, n_dev=aux.plan.view(torch.int32)[1:2], n_mul=16)
        elif ops.sparse_adam_small_ok(rows, self.E):             # (all-positions step: 3*B*S int64 keys, still one launch)
            ops.sparse_adam_rows_small(C, rows.reshape(-1), self.E, self.Em, self.Ev, step=step, lr=self.lr, beta1=b1, beta2=b2, eps=1e-8,
                                       weight_decay=self.wd, padding_idx=0, hyper=hyper)
        elif hyper is not None:
            ops.sparse_adam_rows_dev(C, rows.view(-1).long(), self.E, self.Em, self.Ev, hyper, b1, b2, 1e-8, self.wd, padding_idx=0)
        else:
            ops.sparse_adam_rows(C, rows.view(-1).long(), self.E, self.Em, self.Ev, step, self.lr, b1, b2, 1e-8, self.wd, padding_idx=0)

    def train_step(self, seq, pos, neg, aux=None, grad_hook=None):
        """One step; gradients of the item table exist only as 3*B*S contribution rows."""
        A = self.arena
        if aux is None:
            aux = self.prepare_batch(seq, pos, neg)
        loss, C, rows = self._grads(seq, pos, neg, aux, self._step_seed())
        if grad_hook is not None:
            grad_hook(A.grad)
        A.step += 1
        self._table_adam(C, rows, aux, step=A.step)
        ops.adam_step(A.data, A.grad, A.m, A.v, A.step, self.lr, self.betas[0], self.betas[1], 1e-8, self.wd)
        return loss.squeeze(0)

    # ---- the same step as one hipGraph replay: at D = 128 the block stack is ~300 torch launches per step and the CPU launch
    #      path, not the GPU, sets the step time.  (The torch dropout inside the captured blocks draws from torch's graph-safe
    #      Philox state; the engine's own masks get their per-step seed through the device word, as in SASRecEngine.)
    def _capture(self, B, S, with_adam, blob=None, next_prep=None):
        A = self.arena
        if blob is None:
            blob = torch.zeros(ops.prep_layout(B, S)[1], dtype=torch.uint8, device=self.device)
        state = torch.zeros(4, dtype=torch.int32, device=self.device)
        hyper = state.view(torch.float32)[2:4]
        z = torch.zeros((B, S), dtype=torch.int64, device=self.device)

        def body():
            out = self._grads(pb.seq, pb.pos, pb.neg, pb, 0, seed_dev=state, adam_hyper=hyper if with_adam else None, next_prep=next_prep)
            loss, C, rows = out[:3]
            if with_adam and len(out) == 3:     # (else: both optimizers ran inside the step's two branches)
                self._table_adam(C, rows, pb, hyper=hyper)
                ops.adam_step_dev(A.data, A.grad, A.m, A.v, hyper, self.betas[0], self.betas[1], 1e-8, self.wd)
            return loss, C, rows

        # warm-up on a side stream with an all-padding batch (touches no table row; the arena is restored afterwards)
        keep = [t.clone() for t in (A.data, A.m, A.v, A.grad)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pb = ops.sasrec_batch_prep(z, z, z, blob=blob, state=state, seed=0, step=1, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                                       max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(), weights=self._prep_weights(B, S))
            pb.count.fill_(1)
            for _ in range(3):
                body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with recording(graph, capture_error_mode="thread_local"):
            loss, C, rows = body()
        for t, k in zip((A.data, A.m, A.v, A.grad), keep):
            t.copy_(k)
        return dict(graph=graph, blob=blob, state=state, loss=loss, C=C, rows=rows, pb=pb)

    # ---- the pipelined form (SASRecEngine._train_step_graph_tail): the next batch is prepared by jobs of this step's tail launch
    def _tail_prep_ok(self):
        return bool(getattr(self, "prep_in_tail", True) and self.fused_item_kernel and getattr(self, "fuse_tail", True) and self.encoder == "fused"
                    and self.compact_rows and self.D in (64, 128) and type(self)._grads is SASRecLargeTableEngine._grads
                    and type(self).train_step_graph is SASRecLargeTableEngine.train_step_graph)

    def _tail_pipe(self, B, S):
        if not hasattr(self, "_tail_pipes"):
            self._tail_pipes = {}
        key = (B, S, self.training)
        tp = self._tail_pipes.get(key)
        if tp is None:
            nbytes = ops.prep_layout(B, S)[1]
            blobs = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
            mail = torch.zeros(ops.MAIL_WORDS, dtype=torch.int64, device=self.device)
            graphs = []
            for p in range(2):
                nxt = ops.next_prep(mail, blobs[1 - p], B, S, max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs())
                graphs.append(self._capture(B, S, with_adam=True, blob=blobs[p], next_prep=nxt))
            tp = self._tail_pipes[key] = dict(blobs=blobs, mail=mail, graphs=graphs, parity=0, staged=None)
        return tp

    def _train_step_graph_tail(self, seq, pos, neg, next_batch, next_ready):
        A = self.arena
        B, S = seq.shape
        tp = self._fresh_pipe(self._tail_pipe(B, S))
        p = tp["parity"]
        g = tp["graphs"][p]
        st, tp["staged"] = tp["staged"], None
        if not (st is not None and st[0] is seq and st[1] is pos and st[2] is neg):
            ops.sasrec_batch_prep(seq, pos, neg, blob=tp["blobs"][p], max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs())
        if next_batch is not None and tuple(next_batch[0].shape) != (B, S):
            next_batch = None
        if next_batch is not None and next_ready is not None:
            torch.cuda.current_stream().wait_event(next_ready)
        ops.sasrec_step_stage(g["state"], self._step_seed(), A.step + 1, self.lr, self.betas[0], self.betas[1], B, S, mail=tp["mail"],
                              next_batch=next_batch, weights=self._prep_weights(B, S), loss_acc=self._take_pending_loss())
        g["graph"].replay()
        A.step += 1
        tp["parity"] = 1 - p
        tp["staged"] = next_batch
        self._note_loss(g["loss"], B)
        return g["loss"].squeeze(0)

    def train_step_graph(self, seq, pos, neg, grad_hook=None, next_batch=None, next_ready=None):
        A = self.arena
        B, S = seq.shape
        if grad_hook is None and self._tail_prep_ok() and B <= 8192 and (next_batch is not None or (B, S, self.training) in getattr(self, "_tail_pipes", {})):
            return self._train_step_graph_tail(seq, pos, neg, next_batch, next_ready)
        key = (B, S, grad_hook is None, self.training)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        if key not in self._graphs:
            self._graphs[key] = self._capture(B, S, with_adam=grad_hook is None)
        g = self._graphs[key]
        ops.sasrec_batch_prep(seq, pos, neg, blob=g["blob"], state=g["state"], seed=self._step_seed(), step=A.step + 1, lr=self.lr,
                              beta1=self.betas[0], beta2=self.betas[1], max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(),
                              weights=self._prep_weights(B, S), loss_acc=self._take_pending_loss())
        g["graph"].replay()
        A.step += 1
        if grad_hook is not None:
            grad_hook(A.grad)
            self._table_adam(g["C"], g["rows"], g["pb"], step=A.step)
            ops.adam_step(A.data, A.grad, A.m, A.v, A.step, self.lr, self.betas[0], self.betas[1], 1e-8, self.wd)
        self._note_loss(g["loss"], B)
        return g["loss"].squeeze(0)

    def recommend_topk(self, seq, seen_ptr, seen_idx, K=50):
        u, items = self.encode(seq)
        prep = None if self.training else getattr(self, "_score_prep", None)
        prep = prep[0] if prep is not None and prep[1] == self.arena.step else None
        return ops.score_topk(u[:, -1, :].contiguous(), items, seen_ptr, seen_idx, K, prep=prep)


class SASRecShardedEngine(SASRecLargeTableEngine):
    """The same model with the item table ROW-SHARDED over the process group (BASELINE config 5 at N GPUs: rank r holds rows
    r, r+G, r+2G, ... of the 100 M x 128 table and their Adam moments), everything else replicated and data-parallel:

      * forward: ONE all-to-all round trip fetches the 3*B*S rows a local batch touches (sequence items, positives, negatives)
        into a batch-local table; the embedding front end, the encoder and the criterion run on it unchanged;
      * forward + criterion + backward of every work item in ONE launch on the batch-local table (SASRecEngine's compact-row step);
      * the item-gradient contribution rows (3 per real token, each tagged with the lookup it belongs to) travel to the owners of
        their table rows (one all-to-all) and each owner applies one row-sparse Adam update per distinct row of its shard -- no
        table-sized gradient, no all-reduce of the table; the encoder's dense gradient arena is averaged with one all-reduce
        (`grad_hook` semantics of bench.py).

    With a process group of size 1, dedup=False and the all-positions step (`compact_rows = False`) the step is
    `SASRecLargeTableEngine.train_step` on the same numbers, bit for bit (tests/test_gpu_sasrec.py); the compact-row forms (exact sizes with or
    without dedup -- every distinct row travels once, gradient rows pre-summed per sender --, the fixed-capacity exchange, the captured step)
    give the same sums in another association: the table differs from the unsharded compact-row engine's in rounding only.  The exchange itself
    is covered under gloo with two ranks (tests/test_sharded_gloo.py).  Table values come from `counter_normal_rows`, so every GPU count trains
    the same table.

    capacity_factor c (fixed-capacity form, THE DEFAULT: c = 0.3): every peer pair moves ceil(c * 3 * B * S / G) slots per direction; the
    padding row's lookups take no slot, so c is sized for the REAL tokens (~15 % of a Beauty-shaped batch: c = 0.3 leaves a factor of two).
    Nothing in such a step reads device memory on the host.  A step in which ANY rank's lookups overflowed a bucket is a NO-OP on every
    rank (the ranks learn the global count from one extra word per bucket in the id exchange; both optimizers are gated on it on the device),
    and the host, which reads the count `overflow_lag` steps later, re-runs that batch on the exact-size path (split sizes through the host,
    distinct rows only) with the step number it had -- so an overflow costs a late step, never a wrong one.  `settle_overflow()` drains the
    steps still unchecked (Coach calls it at the end of an epoch).  capacity_factor=None: the exact-size exchange on every step.
    `train_step_graph` needs the fixed-capacity form; call `release_graphs()` before destroying the process group."""

    DEFAULT_CAPACITY_FACTOR = 0.3

    def __init__(self, *args, group=None, dedup=True, capacity_factor="default", local_ops=None, overflow_lag=2, **kw):
        import collections
        import torch.distributed as dist
        self.group = group
        self.dedup = dedup   # exact-size exchange: only the distinct rows of a batch cross the fabric (ShardedTable); False: one row per lookup
        # capacity_factor: the sync-free fixed-capacity exchange (ShardedTable): owner bucketing on the device, equal-split all-to-alls
        self.capacity_factor = self.DEFAULT_CAPACITY_FACTOR if isinstance(capacity_factor, str) else capacity_factor
        self._local_ops = local_ops
        self.overflow_lag, self.overflow_steps = int(overflow_lag), 0          # (steps re-run on the exact path so far)
        self._pending = collections.deque()
        self.world = dist.get_world_size(group)
        kw["table_init"] = "counter"
        super().__init__(*args, **kw)

    def _alloc_table(self, seed):
        from .sharded import ShardedTable
        self.table = ShardedTable(self.N + 1, self.D, group=self.group, device=self.device, dedup=self.dedup,
                                  capacity_factor=self.capacity_factor, local_ops=self._local_ops, skip_row=0)
        T = self.table
        T.raise_on_overflow = False                  # (overflowing steps are gated on the device and re-run here)
        step_rows = max(1, (1 << 24) // self.D)
        for l0 in range(0, T.local_rows, step_rows):
            local = torch.arange(l0, min(T.local_rows, l0 + step_rows), device=self.device)
            T.weight[l0:l0 + step_rows] = counter_normal_rows(T.global_index(local), self.D, seed, self.table_std, self.device)
        if T.rank == 0:
            T.weight[0].zero_()                      # global row 0 = padding
        T.m = torch.zeros_like(T.weight)
        T.v = torch.zeros_like(T.weight)
        self.E = None                                # no rank holds the table

    def state_dict(self, dst=0):
        """Gather-on-save (every rank must call it): the reference's state_dict holds the whole nn.Embedding
        (ETEGRec/train_etegrec.py:549-574); here the shards travel to rank `dst` in row chunks.  -> the full state dict on rank
        `dst` (same keys as SASRecEngine's), the table-less one elsewhere."""
        sd = OrderedDict((k, p.detach().clone()) for k, p in self.params.items())
        full = self.table.gather_full(dst)
        if full is not None:
            sd["Item.embeddings.weight"] = full
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, p in self.params.items():
                p.copy_(torch.as_tensor(sd[k]).to(self.device).view(p.shape))
            if "Item.embeddings.weight" in sd:
                self.table.load_full(sd["Item.embeddings.weight"])

    def optimizer_state(self, dst=0):
        """Adam state incl. the table's moments, gathered like the table (Coach.save_checkpoint)."""
        st = {"m": self.arena.m.clone(), "v": self.arena.v.clone(), "step": self.arena.step}
        Em, Ev = self.table.gather_full(dst, self.table.m), self.table.gather_full(dst, self.table.v)
        if Em is not None:
            st["Em"], st["Ev"] = Em, Ev
        return st

    def load_optimizer_state(self, st):
        self.arena.m.copy_(st["m"]); self.arena.v.copy_(st["v"]); self.arena.step = int(st["step"])
        if "Em" in st:
            self.table.load_full(st["Em"], self.table.m); self.table.load_full(st["Ev"], self.table.v)

    def reset_ranking_buffers(self):
        """Coach.evaluate calls this before a split's batches: nothing to cache (every rank scores its own shard per call)."""

    def _dense_adam(self, hyper=None):
        """hyper (device float32[2]): the step size and bias correction from device memory; {0, 0} = leave everything as it is."""
        A = self.arena
        if hyper is not None:
            ops.adam_step_dev(A.data, A.grad, A.m, A.v, hyper, self.betas[0], self.betas[1], 1e-8, self.wd)
        else:
            ops.adam_step(A.data, A.grad, A.m, A.v, A.step, self.lr, self.betas[0], self.betas[1], 1e-8, self.wd)

    def _hyper(self, step):
        b1, b2 = self.betas
        return torch.tensor([self.lr / (1.0 - b1 ** step), 1.0 / math.sqrt(1.0 - b2 ** step)], dtype=torch.float32).to(self.device, non_blocking=True)

    def _sharded_body(self, seq, pos, neg, aux, sd, grad_hook=None, seed_dev=None, hyper=None, exact=False):
        """Lookup -> batch-local table -> forward + criterion + backward -> gradient rows to their owners -> both optimizers.
        hyper (device float32[2]: step size, bias correction): the captured form; otherwise the host's step count.
        exact: the exact-size exchange for this call (the re-run of a step the fixed-capacity exchange overflowed in).
        Fixed-capacity form: `self._dropped` (device int [1]) = lookups of this step, over ALL ranks, that found no bucket slot; when it is
        not zero both optimizers leave their state untouched (the step is re-run by `_settle`)."""
        import torch.distributed as dist
        A, D = self.arena, self.D
        B, S = seq.shape
        n = B * S
        p = self.p_drop if self.training else 0.0
        slots = positions = gate = None
        fixed = self.capacity_factor is not None and not exact
        if fixed and hyper is None:
            hyper = self._hyper(A.step + 1)        # (the gate below works on the device-side step scalars)
            host_step = True
        else:
            host_step = hyper is None
        if fixed and self.encoder == "fused" and self.compact_rows:
            # fixed-capacity exchange + compact rows: the batch-local table IS the received bucket (row 0 = padding, row 1 + s = bucket
            # slot s); the padding row's lookups were never sent (skip_row), a lookup's local id is its slot + 1, and a contribution
            # row's key - 1 is the bucket slot its gradient travels back in -- nothing is expanded to one row per lookup
            T, route = self.table.lookup(aux.rows_all, expand=False)
            loc = route.slot + 1
            loss, C, keys = self._grads(loc[:n].view(B, S), (loc[n:2 * n] - 1).view(B, S), (loc[2 * n:] - 1).view(B, S), aux, sd,
                                        seed_dev=seed_dev, table=T)
            slots = keys.view(-1).long() - 1 if keys is not None else torch.arange(C.shape[0], device=C.device)   # (None: one row per row of T[1:])
        else:
            # the batch-local table: row 0 = padding, row 1 + j = table row rows_all[j]  (one all-to-all round trip)
            rows, route = self.table.lookup(aux.rows_all, exact=exact)
            T = torch.cat([torch.zeros((1, D), dtype=torch.float32, device=self.device), rows], 0)
            ar = torch.arange(1, n + 1, device=self.device)
            seq_l = torch.where(seq.reshape(-1) != 0, ar, torch.zeros_like(ar)).view(B, S)
            # positives / negatives are rows n+1.. and 2n+1.. of the batch-local table (the criterion adds e_off = 1 to the 0-based ids).
            # compact-row step: C holds 3 x NR contribution rows, `keys` their rows of T (0 = none); key - 1 = the lookup they belong to.
            # (all-positions step, compact_rows = False: one row per lookup, keys None)
            loss, C, keys = self._grads(seq_l, (ar - 1 + n).view(B, S), (ar - 1 + 2 * n).view(B, S), aux, sd, seed_dev=seed_dev, table=T)
            positions = None if keys is None or keys is aux.rows_all else keys.view(-1).long() - 1
        if self.world > 1:
            C.mul_(1.0 / self.world)                 # the loss of the global batch is the mean of the ranks' losses
            dist.all_reduce(A.grad, op=dist.ReduceOp.AVG, group=self.group)
        if grad_hook is not None:
            grad_hook(A.grad)
        if fixed:
            self._dropped = route.dropped
            gate = route.dropped > 0
            hyper = torch.where(gate, torch.zeros_like(hyper), hyper)
        # contribution rows of pad / invalid positions are zero rows addressed to global row 0 (rank 0 drops them)
        self.table.backward_sparse_adam(C, route, A.step + 1, self.lr, self.betas, 1e-8, self.wd, padding_global_row=0, positions=positions,
                                        hyper=hyper, slots=slots, gate=gate)
        if host_step:
            A.step += 1
        self._dense_adam(hyper)
        return loss

    # ---- overflow of the fixed-capacity exchange: the count of step i is read `overflow_lag` steps later (by then the step has long finished:
    #      the host never waits for the device in the steady state), at the same point of the step sequence on every rank -- the re-run's
    #      collectives must line up -- and an overflowed step (a no-op on the device) is re-run on the exact-size path
    def _track(self, batch, step_no, sd):
        if self.capacity_factor is None:
            return
        if self.device.type == "cuda":
            if not hasattr(self, "_ovf_ring"):
                self._ovf_ring = torch.zeros(self.overflow_lag + 2, dtype=self._dropped.dtype).pin_memory()
                self._ovf_i = 0
            k = self._ovf_i % self._ovf_ring.numel()
            self._ovf_i += 1
            self._ovf_ring[k:k + 1].copy_(self._dropped, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((ev, self._ovf_ring[k:k + 1], batch, step_no, sd))
        else:
            self._pending.append((None, self._dropped.clone(), batch, step_no, sd))
        while len(self._pending) > self.overflow_lag:
            self._settle(self._pending.popleft())

    def _settle(self, item):
        ev, word, batch, step_no, sd = item
        if ev is not None:
            ev.synchronize()
        if int(word) == 0:
            return
        self.overflow_steps += 1
        A = self.arena
        now, A.step = A.step, step_no - 1          # the re-run takes the skipped step's number (bias corrections) and dropout seed
        try:
            self._sharded_body(*batch, self.prepare_batch(*batch), sd, exact=True)
        finally:
            A.step = now

    def settle_overflow(self):
        """Check (and re-run where needed) every step not checked yet.  Every rank must call it at the same point."""
        while self._pending:
            self._settle(self._pending.popleft())
        return self.overflow_steps

    def train_step(self, seq, pos, neg, aux=None, grad_hook=None):
        if aux is None:
            aux = self.prepare_batch(seq, pos, neg)
        sd = self._step_seed()
        loss = self._sharded_body(seq, pos, neg, aux, sd, grad_hook=grad_hook).squeeze(0)
        self._track((seq, pos, neg), self.arena.step, sd)
        return loss

    # ---- the same step as ONE hipGraph replay.  Needs the fixed-capacity exchange (capacity_factor: equal-split all-to-alls whose sizes
    #      do not depend on the data -- nothing in the step reads device memory on the host) and RCCL's stream capture; every rank must
    #      capture and replay in step.  Overflowing lookups are counted on the device: `self.table.check_capacity()` at a sync point.
    def _capture(self, B, S, with_adam=True):
        if self.capacity_factor is None:
            raise NotImplementedError("the exchange's split sizes are host-side: train_step_graph needs capacity_factor (fixed-capacity exchange)")
        A = self.arena
        blob = torch.zeros(ops.prep_layout(B, S)[1], dtype=torch.uint8, device=self.device)
        state = torch.zeros(4, dtype=torch.int32, device=self.device)
        hyper = state.view(torch.float32)[2:4]
        z = torch.zeros((B, S), dtype=torch.int64, device=self.device)
        T = self.table
        keep = [t.clone() for t in (A.data, A.m, A.v, A.grad)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):          # warm-up with an all-padding batch: every lookup is the padding row, no table row changes
            pb = ops.sasrec_batch_prep(z, z, z, blob=blob, state=state, seed=0, step=1, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                                       max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(), weights=self._prep_weights(B, S))
            pb.count.fill_(1)
            for _ in range(3):
                self._sharded_body(pb.seq, pb.pos, pb.neg, pb, 0, seed_dev=state, hyper=hyper)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with recording(graph, capture_error_mode="thread_local"):
            loss = self._sharded_body(pb.seq, pb.pos, pb.neg, pb, 0, seed_dev=state, hyper=hyper)
        for t, k in zip((A.data, A.m, A.v, A.grad), keep):
            t.copy_(k)
        if T.dropped is not None:
            T.dropped.zero_()                  # (the all-padding warm-up sends every lookup to the padding row's owner)
        return dict(graph=graph, blob=blob, state=state, loss=loss, dropped=self._dropped)

    def release_graphs(self):
        """Drop the captured steps.  Call before `dist.destroy_process_group()`: a live hipGraph holds the communicator's captured work
        and RCCL's teardown waits for it (measured on RCCL 2.26.6: the destroy call never returns otherwise)."""
        if getattr(self, "_graphs", None):
            self._graphs.clear()
            torch.cuda.synchronize()

    def train_step_graph(self, seq, pos, neg, grad_hook=None):
        if grad_hook is not None:
            raise NotImplementedError("the captured sharded step averages the dense gradients itself (one all-reduce inside the graph)")
        A = self.arena
        B, S = seq.shape
        key = (B, S, self.training)
        if not hasattr(self, "_graphs"):
            self._graphs = {}
        if key not in self._graphs:
            self._graphs[key] = self._capture(B, S)
        g = self._graphs[key]
        ops.sasrec_batch_prep(seq, pos, neg, blob=g["blob"], state=g["state"], seed=self._step_seed(), step=A.step + 1, lr=self.lr,
                              beta1=self.betas[0], beta2=self.betas[1], max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(),
                              weights=self._prep_weights(B, S), loss_acc=self._take_pending_loss())
        sd = self._step_seed()
        g["graph"].replay()
        A.step += 1
        self._dropped = g["dropped"]
        self._track((seq, pos, neg), A.step, sd)
        self._note_loss(g["loss"], B)
        return g["loss"].squeeze(0)

    def encode(self, seq):
        with torch.no_grad():
            B, S = seq.shape
            rows, _ = self.table.lookup(seq.reshape(-1), exact=True)
            T = torch.cat([torch.zeros((1, self.D), dtype=torch.float32, device=self.device), rows], 0)
            ar = torch.arange(1, B * S + 1, device=self.device)
            seq_l = torch.where(seq.reshape(-1) != 0, ar, torch.zeros_like(ar)).view(B, S)
            P, p = self.params, (self.p_drop if self.training else 0.0)
            u, _ = ops.sasrec_embed_encoder_fwd(T, P["Position.weight"].detach(), seq_l, float(self.D ** 0.5), self._block_tensors(),
                                                P["lastLN.weight"].detach(), P["lastLN.bias"].detach(), self.L, p, self._step_seed(),
                                                plan=ops.sasrec_plan(seq_l, self.D))
            return u, None

    def recommend_topk(self, seq, seen_ptr, seen_idx, K=50):
        """Sharded full-catalog top-K (ShardedTable.score_topk): ids are ITEM ids (table row - 1); the padding row never wins."""
        u, _ = self.encode(seq)
        q = u[:, -1, :].contiguous()
        # the table's row 0 is the padding row: mask it as "seen" for every query, and shift the user's seen item ids by one
        b = q.shape[0]
        if seen_ptr is None:
            sp = torch.arange(0, b + 1, device=self.device, dtype=torch.int64)
            si = torch.zeros(b, dtype=torch.int64, device=self.device)
        else:
            cnt = seen_ptr[1:] - seen_ptr[:-1] + 1
            sp = torch.zeros(b + 1, dtype=torch.int64, device=self.device)
            sp[1:] = torch.cumsum(cnt, 0)
            si = torch.empty(int(sp[-1]), dtype=torch.int64, device=self.device)
            si[sp[:-1]] = 0
            body = torch.ones(int(sp[-1]), dtype=torch.bool, device=self.device)
            body[sp[:-1]] = False
            si[body] = seen_idx + 1
        vals, idx = self.table.score_topk(q, sp, si, K)
        return vals, torch.where(idx >= 0, idx - 1, idx)
