"""SASRec on the engine: the host-side mirror of the reference's `SASRec` (SASRec/main.py:53-236) and of the step
loop in `CoachForSASRec.train_per_epoch` (SASRec/main.py:242-258).

Same parameter names/shapes as the reference's state_dict (`Item.embeddings.weight`, `Position.weight`,
`attnLayers.l.in_proj_weight`, `fwdLayers.l.conv1.weight` [D,D,1], ...), same method names (`encode`, `fit`,
`recommend_from_full`) and argument meaning, so the parity tests read like the reference's own code.

MI355X-first layout: every parameter is a view into ONE contiguous fp32 arena in HBM; gradients, and Adam's m/v
live in arenas of the same layout.  Consequences: the optimizer is a single fused launch (re_adam_step), a
data-parallel step needs exactly one RCCL all-reduce (the whole gradient arena is one bucket), and checkpoints
are one tensor.  The arena is padded so every parameter starts 16-byte aligned (float4 access in the kernels).

Hot-path ops are librecengine kernels (recboard_amd/ops.py); there is no CPU fallback.
"""
import math
import os
from collections import OrderedDict

import torch

from . import ops
from .capture import recording
from .lib import load as lib_load


def param_shapes(num_items: int, maxlen: int, D: int, num_blocks: int):
    """Reference state_dict order (SASRec/main.py:70-111)."""
    shapes = OrderedDict()
    shapes["Item.embeddings.weight"] = (num_items + 1, D)  # row 0 = padding (NUM_PADS = 1)
    shapes["Position.weight"] = (maxlen, D)
    for l in range(num_blocks):
        shapes[f"attnLNs.{l}.weight"] = (D,)
        shapes[f"attnLNs.{l}.bias"] = (D,)
    for l in range(num_blocks):
        shapes[f"attnLayers.{l}.in_proj_weight"] = (3 * D, D)
        shapes[f"attnLayers.{l}.in_proj_bias"] = (3 * D,)
        shapes[f"attnLayers.{l}.out_proj.weight"] = (D, D)
        shapes[f"attnLayers.{l}.out_proj.bias"] = (D,)
    for l in range(num_blocks):
        shapes[f"fwdLNs.{l}.weight"] = (D,)
        shapes[f"fwdLNs.{l}.bias"] = (D,)
    for l in range(num_blocks):
        shapes[f"fwdLayers.{l}.conv1.weight"] = (D, D, 1)
        shapes[f"fwdLayers.{l}.conv1.bias"] = (D,)
        shapes[f"fwdLayers.{l}.conv2.weight"] = (D, D, 1)
        shapes[f"fwdLayers.{l}.conv2.bias"] = (D,)
    shapes["lastLN.weight"] = (D,)
    shapes["lastLN.bias"] = (D,)
    return shapes


class ParamArena:
    """Named fp32 views into one contiguous device buffer (+ same-layout buffers for grad / Adam state)."""

    def __init__(self, shapes, device):
        self.shapes = shapes
        self.offsets = OrderedDict()
        off = 0
        for k, shp in shapes.items():
            self.offsets[k] = off
            off += (math.prod(shp) + 3) // 4 * 4
        self.numel = off
        self.data = torch.zeros(off, dtype=torch.float32, device=device)
        self.grad = torch.zeros_like(self.data)
        self.m = torch.zeros_like(self.data)
        self.v = torch.zeros_like(self.data)
        self._step, self.step_gen = 0, 0

    # the optimizer's step count.  The step loops advance it by one; ANY other write (a loaded checkpoint, a reset, a re-run of a skipped step)
    # starts a new `step_gen`: the pipelined step's span hand-over flag (csrc/enc_plan_body.h: PlMail.epoch) is derived from the step number
    # and never reset, so a count that moves back onto a value a staging buffer has already seen would let that buffer's stale flag pass for
    # this step's -- SASRecEngine._fresh_pipe clears the buffers when the generation changed.
    @property
    def step(self):
        return self._step

    @step.setter
    def step(self, n):
        n = int(n)
        if n != self._step + 1:
            self.step_gen += 1
        self._step = n

    def view(self, buf, k):
        o = self.offsets[k]
        return buf[o:o + math.prod(self.shapes[k])].view(self.shapes[k])

    def views(self, buf):
        return OrderedDict((k, self.view(buf, k)) for k in self.shapes)

    def adam_state_dict(self, lr, betas, weight_decay):
        """The optimizer state in torch.optim.Adam's state_dict shape (what the reference's checkpoint.tar holds under "optimizer"):
        parameters numbered in state-dict order, per parameter {step, exp_avg, exp_avg_sq} as copies of the arena's views."""
        names = list(self.shapes)
        state = {i: {"step": torch.tensor(float(self.step)), "exp_avg": self.view(self.m, k).clone(), "exp_avg_sq": self.view(self.v, k).clone()}
                 for i, k in enumerate(names)}
        group = {"lr": lr, "betas": tuple(betas), "eps": 1e-8, "weight_decay": weight_decay, "amsgrad": False, "maximize": False,
                 "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group], "param_names": names}

    def load_adam_state_dict(self, sd):
        if "state" not in sd:                        # (round-1 checkpoints: flat arena copies)
            self.m.copy_(sd["m"]); self.v.copy_(sd["v"]); self.step = int(sd["step"])
            return
        for i, k in enumerate(self.shapes):
            st = sd["state"][i]
            self.view(self.m, k).copy_(st["exp_avg"].to(self.m.device).view(self.shapes[k]))
            self.view(self.v, k).copy_(st["exp_avg_sq"].to(self.v.device).view(self.shapes[k]))
            self.step = int(st["step"])


class SASRecEngine:
    """SASRec (reference defaults: D=64, 2 blocks, 1 head, maxlen 50) with engine kernels on the hot path."""

    def __init__(self, num_items, maxlen=50, embedding_dim=64, num_blocks=2, dropout_rate=0.0, loss="BCE",
                 lr=1e-3, weight_decay=0.0, betas=(0.9, 0.999), device="cuda", seed=1, encoder="fused"):
        assert loss in ("BCE", "BPR", "CE")
        if encoder != "fused":
            raise NotImplementedError("recengine: the engine has ONE encoder, the HIP kernels (a torch restatement of the block stack lives in "
                                      "tests/aten_sasrec.py as a comparator)")
        if embedding_dim not in (64, 128) or maxlen > 64 or num_blocks > 4:
            raise NotImplementedError(f"recengine: unsupported shape (RE_EUNSUPPORTED): the encoder kernels cover D = 64 or 128, maxlen <= 64, "
                                      f"blocks <= 4; got D = {embedding_dim}, maxlen = {maxlen}, blocks = {num_blocks}")
        self.encoder = encoder
        self.split_long = True          # sequences of 3 - 4 tiles as two work items in two workgroups (fused BCE / BPR training step)
        self.fused_item_kernel = True   # forward + criterion + backward of a work item in one launch (False: two launches; same results)
        self.fuse_tail = True           # D = 64: scatter-add and weight-gradient jobs in one launch (csrc/enc_tail.hip; same results)
        self.prep_in_tail = True        # train_step_graph(next_batch=...): the next batch is prepared by jobs of this step's tail launch
        self.fork_wgrad = True          # otherwise: weight gradients on a side stream beside the item table's scatter-add (same results)
        self.tile_step = True           # the one-tile-per-workgroup kernels may run the training step (False: always the workgroup-per-item kernels)
        self.fuse_adam = True           # captured steps: the dense Adam inside the launches that finish the gradients (reduction / scatter-add)
        self.fork_adam = False          # True: the dense Adam as two launches inside the step's two branches (measured: 113 vs 104 us per step)
        self.pipelined_prep = False     # True: the Coach hands train_step_graph the next batch and its preparation launch runs on a side stream
                                        # beside this step.  Measured on MI355X: 115 - 137 us per step against 106 in front of the step -- the
                                        # second queue's launch disturbs the graph's own branches more than the 12 us it hides
        self.ce_logits_bytes = 1 << 28  # loss='CE': at most this many bytes of logits at a time (more: the catalog is walked in chunks)
        self.compact_rows = True     # BCE / BPR fused step on the batch plan's compact rows (False: all B*S positions + sorted scatter-add)
        self._bufs = {}
        self.N, self.S, self.D, self.L = num_items, maxlen, embedding_dim, num_blocks
        self.p_drop, self.loss_kind = dropout_rate, loss
        self.lr, self.wd, self.betas = lr, weight_decay, betas
        self.device = torch.device(device)
        self.arena = ParamArena(param_shapes(num_items, maxlen, embedding_dim, num_blocks), self.device)
        self.training = True
        self.seed = seed
        self.params = OrderedDict()
        for k in self.arena.shapes:
            p = self.arena.view(self.arena.data, k).requires_grad_(True)
            self.params[k] = p
        self.reset_parameters(seed)

    # ---- reference SASRec.reset_parameters (SASRec/main.py:130-141): embeddings + nn.Linear (= out_proj) xavier-normal,
    #      MHA in_proj xavier-uniform, Conv1d default kaiming-uniform(a=sqrt(5)), LN gamma=1 beta=0, biases 0
    def reset_parameters(self, seed=1):
        g = torch.Generator(device="cpu").manual_seed(seed)
        D = self.D
        with torch.no_grad():
            for k, p in self.params.items():
                shp = tuple(p.shape)
                if k.endswith("embeddings.weight") or k == "Position.weight" or k.endswith("out_proj.weight"):
                    std = math.sqrt(2.0 / (shp[0] + shp[1]))
                    w = torch.randn(shp, generator=g) * std
                elif k.endswith("in_proj_weight"):
                    bound = math.sqrt(6.0 / (shp[0] + shp[1]))
                    w = (torch.rand(shp, generator=g) * 2 - 1) * bound
                elif "conv" in k:
                    bound = 1.0 / math.sqrt(D)
                    w = (torch.rand(shp, generator=g) * 2 - 1) * bound
                elif "LN" in k and k.endswith("weight"):
                    w = torch.ones(shp)
                else:
                    w = torch.zeros(shp)
                p.copy_(w.to(self.device))

    def load_state_dict(self, sd):
        with torch.no_grad():
            for k, p in self.params.items():
                p.copy_(torch.as_tensor(sd[k]).to(self.device).view(p.shape))
        self._score_prep = None

    def state_dict(self):
        return OrderedDict((k, p.detach().clone()) for k, p in self.params.items())

    def train(self, mode=True):
        self.training = mode
        if mode:
            self._score_prep = None   # the item table is about to change
        return self

    def reset_ranking_buffers(self):
        """Called by Coach.evaluate before a split's batches (freerec contract; MF-BPR/main.py:95-99 clones its tables here):
        split the item table once for all of the split's `recommend_topk` calls (re_score_prepare)."""
        self._score_prep = (ops.score_prepare(self.params["Item.embeddings.weight"].detach()[1:]), self.arena.step)

    def eval(self):
        return self.train(False)

    def _block_tensors(self, buf=None):
        A = self.arena
        named = self.params if buf is None else A.views(buf)
        return ops.sasrec_block_tensors({k: v.detach() for k, v in named.items()}, self.L)

    def _step_seed(self):
        """32-bit dropout seed of the current step (counter-based masks: (seed, stream, element) -> keep bit)."""
        return (self.seed * 0x9E3779B1 + (self.arena.step + 1) * 0x85EBCA77) & 0xFFFFFFFF

    def encode(self, seq):
        """-> (userEmbds [B,S,D], itemEmbds = E[1:]).  SASRec/main.py:178-193.  Forward only (training goes through `train_step*`)."""
        P = self.params
        E = P["Item.embeddings.weight"].detach()
        p = self.p_drop if self.training else 0.0
        sd = self._step_seed()
        u, _ = ops.sasrec_embed_encoder_fwd(E, P["Position.weight"].detach(), seq, float(self.D ** 0.5), self._block_tensors(),
                                            P["lastLN.weight"].detach(), P["lastLN.bias"].detach(), self.L, p, sd,
                                            plan=ops.sasrec_batch_prep(seq, max_tiles=self._max_tiles()).plan)
        return u, E[1:]

    def fit(self, seq, pos, neg, aux=None):
        """-> {"rec_loss": scalar}, forward only.  SASRec/main.py:195-221 (BCE / BPR over the non-pad positions; CE over the catalog)."""
        u, items = self.encode(seq)
        valid, _, _ = aux if aux is not None else self.batch_aux(seq, pos, neg)
        u2 = u.reshape(-1, self.D)
        if self.loss_kind == "CE":
            vidx = torch.nonzero(valid).reshape(-1)
            logits = ops.gemm(ops.gather_rows(u2, vidx), items, transB=True)
            return {"rec_loss": ops.ce_rows_(logits, pos.reshape(-1)[vidx].contiguous()).squeeze(0)}
        kind = ops.LOSS_BCE if self.loss_kind == "BCE" else ops.LOSS_BPR
        loss, _, _ = ops.pair_loss_fwd(u2, self.params["Item.embeddings.weight"].detach(), pos.reshape(-1), neg.reshape(-1), valid, kind, e_off=1)
        return {"rec_loss": loss.squeeze(0)}

    @staticmethod
    def batch_aux(seq, pos, neg):
        """Per-batch index helpers (belong to batch assembly, not to the step): valid mask as uint8 and the
        destination rows of the pos/neg gradient contributions (row 0 = padding row, dropped)."""
        v = (seq != 0).reshape(-1)
        return (v.to(torch.uint8), torch.where(v, pos.reshape(-1) + 1, 0), torch.where(v, neg.reshape(-1) + 1, 0))

    def recommend_from_full(self, seq):
        """scores [B, N] = u[:, -1, :] . E[1:]^T  (SASRec/main.py:223-228) -- dense drop-in."""
        with torch.no_grad():
            u, items = self.encode(seq)
            return ops.score_dense(u[:, -1, :].contiguous(), items)

    def recommend_topk(self, seq, seen_ptr, seen_idx, K=50):
        """Coach.evaluate contract fused (UniSRec/main.py:408-414): masked top-K without the B x N matrix."""
        with torch.no_grad():
            u, items = self.encode(seq)
            prep = None if self.training else getattr(self, "_score_prep", None)
            prep = prep[0] if prep is not None and prep[1] == self.arena.step else None   # (planes of the table as it is NOW)
            return ops.score_topk(u[:, -1, :].contiguous(), items, seen_ptr, seen_idx, K, prep=prep)

    def recommend_from_pool(self, seq, pool):
        """scores [B, P] of every sequence's candidate pool (`pool` int64 [B, P], 0-based item ids: the target first, then the sampled
        unseen items): SASRec/main.py:230-236, einsum("BD,BKD->BK") on the last position -- one gather-and-dot launch (re_score_pool)."""
        with torch.no_grad():
            u, items = self.encode(seq)
            return ops.score_pool(u[:, -1, :].contiguous(), items, pool.contiguous())

    def _max_tiles(self):
        return 4 if self.D == 64 else 2     # tiles of 16 rows per work item (LDS capacity of the workgroup-per-item encoder kernels)

    def prepare_batch(self, seq, pos, neg, for_next_step=False):
        """Per-batch preparation of the fused step as ONE engine launch (re_sasrec_batch_prep; what the reference does at the top of
        `fit`, SASRec/main.py:199-204, plus the encoder's work plan): valid mask, number of valid positions, destination rows of the
        3*B*S gradient contributions, work items.  No host sync.  -> ops.PreparedBatch.
        for_next_step: the launch also prepares the weights of the step (re_sasrec_batch_prep_w) -- the batch must then go into the very
        next step, before anything else changes the parameters."""
        return ops.sasrec_batch_prep(seq, pos, neg, max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(), ncu=self._plan_ncu(),
                                     weights=self._prep_weights(*seq.shape) if for_next_step else None)

    def check_handover(self):
        """Raise if a split sequence's halves ever timed out waiting for each other (Coach calls this once per epoch; host sync)."""
        for (B, S), W in self._bufs.items():
            err = ops.sasrec_tape_errors(W["tape"], B, S)
            if err:
                ops.sasrec_tape_reset_flags(W["tape"], B, S)    # (a late producer store must not read as "published" in the next replay)
                raise RuntimeError("recengine: a split-sequence plan was launched on fewer workgroups than it has work items (the plan's and the "
                                   "step's workgroup counts disagree); that step did nothing" if err == 2 else
                                   "recengine: a split sequence's work items did not meet (hand-over time-out); results of that step are invalid")

    def _wave_step(self):
        """The training step may run one tile per workgroup (csrc/enc_tile.hip: D = 64 and 128; re_sasrec_encoder_step picks per batch)."""
        ok = bool(self.D in (64, 128) and getattr(self, "tile_step", True) and self.fused_item_kernel and self.encoder == "fused" and self.loss_kind != "CE" and self.compact_rows)
        return "always" if ok and getattr(self, "tile_step", True) == "always" else ok   # ("always": whatever the batch -- the plan's rule is a matter of speed)

    def _tile_wgs(self):
        """Resident workgroups per CU of the tile kernels, as the LOADED library was built (csrc/enc_common.h: enc_tile_wg_per_cu -- two at D = 64
        since round 6, one at D = 128; `make one` builds the one-per-CU library): the batch plan's residency rule has to count the same number,
        so the host asks instead of assuming.  RE_TILE_WGS in the environment is checked against it (older scripts set it)."""
        have = int(lib_load().re_tile_wgs_per_cu(self.D))
        want = os.environ.get("RE_TILE_WGS")
        if want is not None and int(want) != have:
            raise RuntimeError(f"recengine: RE_TILE_WGS={want} but the loaded library holds {have} tile workgroup(s) per CU (make one builds the other)")
        return have

    def _tail_word(self):
        if not hasattr(self, "_tail"):
            self._tail = torch.zeros(4, dtype=torch.int32, device=self.device)
        return self._tail

    def _prep_weights(self, B, S):
        """What the batch preparation launch needs to also prepare the tile step's weight fragments (ops.sasrec_batch_prep(weights=));
        None when the step that follows is not the one-tile-per-workgroup one."""
        if not (self._wave_step() and self.training and getattr(self, "prep_weights_in_batch_prep", True)):
            return None
        cache = self.__dict__.setdefault("_pw_cache", {})
        mw = cache.get((B, S))
        if mw is None or mw.keep[4] is not self._buffers(B, S)["tape"] or mw.keep[1].data_ptr() != self.params["lastLN.weight"].data_ptr():
            W = self._buffers(B, S)
            P = self.params
            mw = cache[(B, S)] = ops.marshal_weights((self._block_tensors(), P["lastLN.weight"].detach(), P["lastLN.bias"].detach(), self.L,
                                                     W["tape"], W["ws_bwd"]))
        return mw

    def _plan_ncu(self):
        """Workgroups the batch plan's items should fill (None: the device's CUs).  The one-tile-per-workgroup step does not read the
        items at all; its fallback, the workgroup-per-item kernel, wants the default."""
        return None

    def _split(self):
        """Long sequences as two work items in two workgroups: the workgroup-per-item form of the fused training step (its tape carries
        the hand-over flags).  The wave-per-tile step gives every tile a wave of its own instead."""
        return bool(self.split_long and self.encoder == "fused" and self.loss_kind != "CE" and self.compact_rows and not self._wave_step())

    def _buffers(self, B, S):
        key = (B, S)
        if key not in self._bufs:
            D, dev, L = self.D, self.device, lib_load()
            f = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)  # noqa: E731
            u8 = lambda n: torch.empty(max(int(n), 256), dtype=torch.uint8, device=dev)  # noqa: E731
            n3 = 3 * B * S
            NR = ops.sasrec_plan_rows(B, S)
            self._bufs[key] = dict(
                u=f(B, S, D), dU=f(B * S, D), contrib=f(n3, D),
                # the step on the plan's compact rows (BCE / BPR): upstream gradient rows, the three contribution-row sets, their keys
                dU_rows=f(NR, D), g_rows=f(3, NR, D), keys=torch.zeros((3, NR), dtype=torch.int32, device=dev),
                ws_loss=torch.zeros(L.re_sasrec_loss_rows_workspace_bytes(), dtype=torch.uint8, device=dev),
                tape=torch.zeros(L.re_sasrec_tape_bytes(B, S, D, self.L) // 4, dtype=torch.float32, device=dev),   # (zero: the hand-over flags)
                ws_bwd=u8(L.re_sasrec_encoder_bwd_workspace_bytes(B, S, D, self.L)),
                ws_sc=u8(L.re_scatter_add_rows_workspace_bytes(n3, D, self.N + 1)))
        return self._bufs[key]

    def _step_body(self, pb, sd, seed_dev=None, adam_hyper=None, next_prep=None):
        """Every launch of the fused step after the batch preparation up to (not including) the optimizer; gradients land in the
        gradient arena.  pb: ops.PreparedBatch.
        adam_hyper (device float32[2], captured steps with their own optimizer): the dense Adam too -- in the two-branch form as TWO
        launches, the item table's slice of the arena behind the scatter-add on one branch and the encoder's slice behind the weight
        gradients on the other, instead of one launch behind the join.  -> (loss, True) then."""
        A, P, D = self.arena, self.params, self.D
        seq, pos, neg = pb.seq, pb.pos, pb.neg
        B, S = seq.shape
        W = self._buffers(B, S)
        G = A.views(A.grad)
        p = self.p_drop if self.training else 0.0
        E, Ppos = P["Item.embeddings.weight"].detach(), P["Position.weight"].detach()
        lw, lb = P["lastLN.weight"].detach(), P["lastLN.bias"].detach()
        bt = self._block_tensors()
        kind = ops.LOSS_BCE if self.loss_kind == "BCE" else ops.LOSS_BPR
        n = B * S
        GE = G["Item.embeddings.weight"]
        if self.loss_kind != "CE" and self.compact_rows:
            # forward + criterion (one launch), encoder backward, weight gradients, item-table gradient -- all on the plan's compact
            # rows: only rows that exist are read or written, and the table gradient is ONE launch over ~13 k keys (no sort)
            ready = 8 if getattr(pb, "weights_ready", False) else 0
            if self.fused_item_kernel and D == 64 and getattr(self, "fuse_tail", True):
                # ONE queue: item kernels, then one launch in which the scatter-add's workgroups (+ the item table's Adam) go on with the
                # weight-gradient jobs, then the reduction (+ the encoder slice's Adam) -- csrc/enc_tail.hip.  (As two graph branches the
                # fork and the join cost more than half of what the overlap saved.)
                loss = ops.sasrec_encoder_step(E, Ppos, seq, pos, neg, float(D ** 0.5), bt, lw, lb, self.L, p, sd, pb.plan, kind, pb.count, W["u"], W["tape"],
                                               W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"], W["contrib"][:n].view(B, S, D), G["Position.weight"],
                                               self._block_tensors(A.grad), G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"], e_off=1, seed_dev=seed_dev,
                                               part=(1 if ops.tile_step_certain(B, S, D, self._split(), self._wave_step(), self._tile_wgs()) else 3) + ready)
                fuse = adam_hyper is not None and getattr(self, "fuse_adam", True)
                fz = ops.adam_fuse(A.grad, A.data, A.m, A.v, adam_hyper, self.betas[0], self.betas[1], 1e-8, self.wd) if fuse else None
                self._adam_keep = (fz,)
                if not hasattr(self, "_ticket"):
                    self._ticket = torch.zeros(1, dtype=torch.int32, device=self.device)
                ops.sasrec_step_tail(W["g_rows"], W["keys"], self.N + 1, GE if (not fuse or getattr(self, "keep_table_grad", True)) else None,
                                     pb.plan.view(torch.int32)[1:2], 16, seq, self.L, pb.plan, W["tape"], W["contrib"][:n].view(B, S, D), float(D ** 0.5),
                                     G["Position.weight"], self._block_tensors(A.grad), G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"], self._ticket,
                                     table_adam=fz, enc_adam=fz, next=next_prep)
                return (loss, True) if fuse else loss
            if self.fused_item_kernel and getattr(self, "fork_wgrad", False):
                # the item kernels, then TWO branches: the weight gradients (enc_wgrad_k + enc_grad_reduce_k) on a side stream beside the
                # item table's scatter-add on this one -- both depend on the item kernels alone; joined before the optimizer (inside a
                # captured step the branches are parallel paths of the graph).  (The two item kernels as parallel branches too -- one of
                # them returns at once -- was measured: 128 vs 110 us per step; a branch at the head of the graph costs more than it hides.)
                args = (E, Ppos, seq, pos, neg, float(D ** 0.5), bt, lw, lb, self.L, p, sd, pb.plan, kind, pb.count, W["u"], W["tape"], W["dU_rows"],
                        W["g_rows"], W["keys"], W["ws_loss"], W["contrib"][:n].view(B, S, D), G["Position.weight"], self._block_tensors(A.grad),
                        G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"])
                main = torch.cuda.current_stream()
                if not hasattr(self, "_side"):
                    self._side = torch.cuda.Stream()
                side = self._side
                loss = ops.sasrec_encoder_step(*args, e_off=1, seed_dev=seed_dev, part=3 + ready)
                side.wait_stream(main)
                ne = A.offsets["Position.weight"]          # the arena's first slice is the item table
                b1, b2 = self.betas
                fuse = adam_hyper is not None and getattr(self, "fuse_adam", True)
                if fuse:
                    # the optimizer rides in the two launches that FINISH the gradients: the reduction applies the encoder slice's dense
                    # Adam, the scatter-add's row owners the item table's -- no optimizer launch behind the join
                    enc_adam = ops.adam_fuse(A.grad, A.data, A.m, A.v, adam_hyper, b1, b2, 1e-8, self.wd)
                    tab_adam = ops.adam_fuse(A.grad, A.data, A.m, A.v, adam_hyper, b1, b2, 1e-8, self.wd)   # (rows [0, N + 1) of the arenas)
                    self._adam_keep = (enc_adam, tab_adam)
                with torch.cuda.stream(side):
                    ops.sasrec_encoder_step(*args, e_off=1, seed_dev=seed_dev, part=4, loss=loss, adam=enc_adam if fuse else None)
                    if adam_hyper is not None and not fuse and getattr(self, "fork_adam", False):
                        ops.adam_step_dev(A.data[ne:], A.grad[ne:], A.m[ne:], A.v[ne:], adam_hyper, b1, b2, 1e-8, self.wd)
                # (the table gradient is still written in the fused form -- arena.grad stays the step's gradient -- unless keep_table_grad is off)
                ops.scatter_add_rows_small(W["g_rows"], W["keys"], self.N + 1, GE if (not fuse or getattr(self, "keep_table_grad", True)) else None,
                                           n_regions=3, n_dev=pb.plan.view(torch.int32)[1:2], n_mul=16, adam=tab_adam if fuse else None)
                if adam_hyper is not None and not fuse and getattr(self, "fork_adam", False):
                    ops.adam_step_dev(A.data[:ne], A.grad[:ne], A.m[:ne], A.v[:ne], adam_hyper, b1, b2, 1e-8, self.wd)
                main.wait_stream(side)
                if fuse or (adam_hyper is not None and getattr(self, "fork_adam", False)):
                    if getattr(self, "tail_node", True):
                        ops.step_state(self._tail_word(), 0, 1, 1e-3)     # (a graph that ENDS in a join of two branches replays slower: one trivial node behind it)
                    return loss, True
                if adam_hyper is None and seed_dev is not None:           # (captured without its optimizer -- the data-parallel form: the graph would end in the join)
                    ops.step_state(self._tail_word(), 0, 1, 1e-3)
                return loss
            if self.fused_item_kernel and adam_hyper is not None and getattr(self, "fuse_adam", True):
                # one queue: item kernels -> weight gradients -> reduction (+ the encoder slice's Adam) -> scatter-add (+ the table's Adam)
                b1, b2 = self.betas
                fz = ops.adam_fuse(A.grad, A.data, A.m, A.v, adam_hyper, b1, b2, 1e-8, self.wd)
                self._adam_keep = (fz,)
                loss = ops.sasrec_encoder_step(E, Ppos, seq, pos, neg, float(D ** 0.5), bt, lw, lb, self.L, p, sd, pb.plan, kind, pb.count,
                                               W["u"], W["tape"], W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"],
                                               W["contrib"][:n].view(B, S, D), G["Position.weight"], self._block_tensors(A.grad),
                                               G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"], e_off=1, seed_dev=seed_dev, part=7 + ready, adam=fz)
                ops.scatter_add_rows_small(W["g_rows"], W["keys"], self.N + 1, GE if getattr(self, "keep_table_grad", True) else None,
                                           n_regions=3, n_dev=pb.plan.view(torch.int32)[1:2], n_mul=16, adam=fz)
                return loss, True
            if self.fused_item_kernel:
                loss = ops.sasrec_encoder_step(E, Ppos, seq, pos, neg, float(D ** 0.5), bt, lw, lb, self.L, p, sd, pb.plan, kind, pb.count,
                                               W["u"], W["tape"], W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"],
                                               W["contrib"][:n].view(B, S, D), G["Position.weight"], self._block_tensors(A.grad),
                                               G["lastLN.weight"], G["lastLN.bias"], W["ws_bwd"], e_off=1, seed_dev=seed_dev, part=ready)
            else:
                loss = ops.sasrec_encoder_fwd_loss(E, Ppos, seq, pos, neg, float(D ** 0.5), bt, lw, lb, self.L, p, sd, pb.plan, kind,
                                                   pb.count, W["u"], W["tape"], W["dU_rows"], W["g_rows"], W["keys"], W["ws_loss"], e_off=1,
                                                   seed_dev=seed_dev)
                ops.sasrec_encoder_bwd(None, seq, bt, lw, lb, self.L, p, sd, W["tape"], self._block_tensors(A.grad), G["lastLN.weight"],
                                       G["lastLN.bias"], out=W["contrib"][:n].view(B, S, D), ws=W["ws_bwd"], plan=pb.plan, seed_dev=seed_dev,
                                       embed_scale=float(D ** 0.5), dP=G["Position.weight"], dU_rows=W["dU_rows"],
                                       out_rows=W["g_rows"][0])
            ops.scatter_add_rows_small(W["g_rows"], W["keys"], self.N + 1, GE, n_regions=3, n_dev=pb.plan.view(torch.int32)[1:2], n_mul=16)
            return loss
        ops.sasrec_embed_encoder_fwd(E, Ppos, seq, float(D ** 0.5), bt, lw, lb, self.L, p, sd, need_tape=True, out=W["u"], tape=W["tape"],
                                     plan=pb.plan, seed_dev=seed_dev)
        u2 = W["u"].view(n, D)
        posf, negf = pos.reshape(-1), neg.reshape(-1)
        C = W["contrib"]
        if self.loss_kind == "CE":
            # SASRec/main.py:216-219: logits = u[valid] E[1:]^T, mean CE against IPos -- fp32 MFMA GEMMs + row kernels.  The catalog is
            # walked in column chunks of at most `ce_logits_bytes` of logits: one chunk (the usual case: 190 MB at Beauty's shapes, 0.07 %
            # of the HBM) is the materialised form; more chunks recompute each chunk's logits in the backward instead of keeping M x N.
            vidx = torch.nonzero(pb.valid).reshape(-1).contiguous()          # (host sync: the CE shapes depend on the batch)
            Uv = ops.gather_rows(u2, vidx)                                   # [M, D]
            M, Nitems = vidx.numel(), self.N
            labels = posf[vidx].contiguous()
            Nc = max(1, min(Nitems, self.ce_logits_bytes // (4 * max(M, 1))))
            if Nc >= Nitems:
                logits = ops.gemm(Uv, E[1:], transB=True)                    # [M, N]
                loss = ops.ce_rows_(logits, labels)                          # logits <- d loss / d logits
                dUv = ops.gemm(logits, E[1:])                                # [M, D]
                chunks = [(0, logits)]
            else:
                st = ops.CEStats(M, self.device)
                buf = torch.empty((M, Nc), dtype=torch.float32, device=self.device)
                for c0 in range(0, Nitems, Nc):
                    lc = ops.gemm(Uv, E[1 + c0:1 + min(c0 + Nc, Nitems)], transB=True, out=buf[:, :min(Nc, Nitems - c0)])
                    ops.ce_chunk_stats(lc, c0, labels, st)
                loss = ops.ce_chunk_loss(st, labels, Nitems)
                # backward, chunk by chunk: logits recomputed, turned into their gradient in place, folded into dU and into the
                # chunk's rows of the table gradient (the scatter-add below then ACCUMULATES the encoder's contribution rows onto them)
                dUv = torch.zeros((M, D), dtype=torch.float32, device=self.device)
                GE[0].zero_()
                for c0 in range(0, Nitems, Nc):
                    c1 = min(c0 + Nc, Nitems)
                    lc = ops.gemm(Uv, E[1 + c0:1 + c1], transB=True, out=buf[:, :c1 - c0])
                    ops.ce_chunk_grad_(lc, c0, labels, st)
                    ops.gemm(lc, E[1 + c0:1 + c1], beta=1.0, out=dUv)
                    ops.gemm(lc, Uv, transA=True, out=GE[1 + c0:1 + c1])
                chunks = None
            ops.scatter_add_rows(dUv, vidx, n, out=W["dU"])                  # back to the [B*S, D] layout (pads zero)
            C[n:].zero_()                                                    # no pos/neg contribution rows in CE mode
        else:   # (compact_rows = False: the criterion over all B*S positions and the sorted scatter-add -- what the large-table engines run)
            loss, _, _, _ = ops.pair_loss_fwd_bwd(u2, E, posf, negf, pb.valid, kind, pb.count, e_off=1, out=(W["dU"], C[n:2 * n], C[2 * n:]))
        ops.sasrec_encoder_embed_bwd(W["dU"].view(B, S, D), seq, float(D ** 0.5), bt, lw, lb, self.L, p, sd, W["tape"],
                                     self._block_tensors(A.grad), G["lastLN.weight"], G["lastLN.bias"], G["Position.weight"],
                                     out=C[:n].view(B, S, D), ws=W["ws_bwd"], plan=pb.plan, seed_dev=seed_dev)
        chunked = self.loss_kind == "CE" and chunks is None
        ops.scatter_add_rows(C, pb.rows_all, self.N + 1, 0, 1.0, out=GE, ws=W["ws_sc"], accumulate=chunked)
        if self.loss_kind == "CE" and not chunked:
            ops.gemm(logits, Uv, transA=True, beta=1.0, out=GE[1:])          # dE[1:] += dlogits^T u[valid]
        return loss

    def train_step_fused(self, seq, pos, neg, aux=None, grad_hook=None):
        """One training step with every hot-path op a librecengine kernel and no autograd graph:
        batch prep -> embed + fused encoder (tape) -> fused pair loss (+ backward) -> encoder backward (all blocks, embedding
        backward fused in) -> weight gradients + reduction -> ONE deterministic scatter-add of all 3*B*S item-gradient rows -> fused Adam.
        aux: an ops.PreparedBatch of this batch (prepare_batch), or None."""
        A = self.arena
        pb = aux if aux is not None else self.prepare_batch(seq, pos, neg, for_next_step=True)
        loss = self._step_body(pb, self._step_seed())
        A.step += 1
        self._hook_and_adam(grad_hook)
        return loss.squeeze(0)

    # ---- the same step as ONE hipGraph replay (the step is ~15 short launches: at B=512 the CPU launch path, not the GPU,
    #      sets the step time).  Per step: the batch-preparation launch (raw (seq, pos, neg) -> the static buffers the graph reads:
    #      copies, valid / count / rows_all, the encoder's plan, per-step seed and Adam scalars as device words) + one graph launch.
    #      BCE / BPR only: the CE path's shapes depend on the batch's number of valid positions.
    @staticmethod
    def _sync_kind(grad_hook):
        """How a captured step ends: True = the fused / dense Adam inside the graph (one replica); the hook itself = a data-parallel step that
        can be recorded (OwnerAdam: collectives + the owner's launch, inside the graph); False = gradients only (a plain hook, then Adam,
        behind the replay)."""
        if grad_hook is None:
            return True
        if getattr(grad_hook, "owns_adam", False) and getattr(grad_hook, "in_graph", True):
            return grad_hook
        return False

    def _capture(self, B, S, with_adam, in_prep=True, blob=None, next_prep=None):
        """with_adam: see _sync_kind.  in_prep: the tile step's weight fragments come from the batch-preparation launch (False: from a launch inside the graph --
        the pipelined form, whose preparation launch runs before the previous step's optimizer has finished)."""
        A = self.arena
        if blob is None:
            blob = torch.zeros(ops.prep_layout(B, S)[1], dtype=torch.uint8, device=self.device)
        state = torch.zeros(4, dtype=torch.int32, device=self.device)
        hyper = state.view(torch.float32)[2:4]
        z = torch.zeros((B, S), dtype=torch.int64, device=self.device)

        def body():
            loss = self._step_body(pb, 0, seed_dev=state, adam_hyper=hyper if with_adam is True else None, next_prep=next_prep)
            if isinstance(loss, tuple):          # (the optimizer ran inside the step's two branches)
                return loss[0]
            if with_adam is True:
                ops.adam_step_dev(A.data, A.grad, A.m, A.v, hyper, self.betas[0], self.betas[1], 1e-8, self.wd)
            elif with_adam:                      # a recboard_amd.dp.OwnerAdam: the data-parallel step's two collectives + the owner's launch, recorded too
                with_adam.step(A.data, A.grad, A.m, A.v, 0, 0.0, self.betas, 1e-8, self.wd, hyper=hyper)
            return loss

        # warm-up on a side stream (workspace allocation, lazy module loads) with an all-padding batch, then restore
        # everything the warm-up touched; the capture itself only records.
        keep = [t.clone() for t in (A.data, A.m, A.v, A.grad)]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            pb = ops.sasrec_batch_prep(z, z, z, blob=blob, state=state, seed=0, step=1, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                                       max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(), ncu=self._plan_ncu(),
                                       weights=self._prep_weights(B, S) if in_prep else None)
            body()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with recording(graph, capture_error_mode="thread_local"):
            loss = body()
        for t, k in zip((A.data, A.m, A.v, A.grad), keep):
            t.copy_(k)
        return dict(graph=graph, blob=blob, state=state, loss=loss, in_prep=in_prep, pb=pb, hyper=hyper)

    def _stage(self, g, seq, pos, neg, step):
        """The batch-preparation launch of `step` (1-based) into the static buffers of the captured step `g`."""
        B, S = seq.shape
        sd = (self.seed * 0x9E3779B1 + step * 0x85EBCA77) & 0xFFFFFFFF          # (= _step_seed() once arena.step == step - 1)
        ops.sasrec_batch_prep(seq, pos, neg, blob=g["blob"], state=g["state"], seed=sd, step=step, lr=self.lr,
                              beta1=self.betas[0], beta2=self.betas[1], max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(), ncu=self._plan_ncu(),
                              weights=self._prep_weights(B, S) if g["in_prep"] else None, loss_acc=self._take_pending_loss())

    # ---- the epoch's loss sum without a launch per step: between begin_ and end_loss_accumulation every captured step's loss is added
    #      (times its batch size) into one device word by the NEXT step's batch-preparation launch; the last one by end_.
    def begin_loss_accumulation(self):
        self._loss_acc = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._loss_pending = None
        return self._loss_acc

    def _take_pending_loss(self):
        acc = getattr(self, "_loss_acc", None)
        pend = getattr(self, "_loss_pending", None)
        if acc is None or pend is None:
            return None
        self._loss_pending = None
        return (pend[0], acc, pend[1])

    def _note_loss(self, loss, weight):
        if getattr(self, "_loss_acc", None) is not None:
            if self._loss_pending is not None:           # (a step whose preparation launch could not carry it: the pipelined form)
                self._loss_acc.add_(self._loss_pending[0], alpha=self._loss_pending[1])
            self._loss_pending = (loss, float(weight))

    def end_loss_accumulation(self):
        """-> the device word holding sum(loss_i * batch size_i) since begin_loss_accumulation."""
        acc, self._loss_acc = self._loss_acc, None
        if self._loss_pending is not None:
            acc.add_(self._loss_pending[0], alpha=self._loss_pending[1])
        self._loss_pending = None
        return acc

    def _hook_and_adam(self, grad_hook):
        """The optimizer step behind a gradient hook (arena.step already advanced).  A hook with `owns_adam` (recboard_amd.dp.OwnerAdam: the
        data-parallel step whose Adam runs on each slice's owner) performs the update itself; a plain hook (e.g. an all-reduce of the gradient
        arena) is followed by the dense Adam launch."""
        A = self.arena
        if grad_hook is not None and getattr(grad_hook, "owns_adam", False):
            grad_hook.step_arena(A, self.lr, self.betas, 1e-8, self.wd)
            return
        if grad_hook is not None:
            grad_hook(A.grad)
        ops.adam_step(A.data, A.grad, A.m, A.v, A.step, self.lr, self.betas[0], self.betas[1], 1e-8, self.wd)

    def train_step_graph(self, seq, pos, neg, grad_hook=None, next_batch=None, next_ready=None):
        """`train_step_fused` on a RAW batch, replayed from a captured hipGraph: one batch-preparation launch (which also stages the
        batch and the step scalars into the graph's static buffers) + one graph launch.  Results are identical to the eager fused
        step.  The returned loss tensor is overwritten by the next call (the call after next with `next_batch`).
        next_batch = (seq, pos, neg) of the FOLLOWING call, if the caller already has it (an epoch loop does): it is prepared DURING this
        step instead of in front of the next one -- the preparation depends on the batch alone -- into the buffers of a second captured
        copy of the step (two copies alternate): by jobs of this step's tail launch (`prep_in_tail`, D = 64 fused steps: _train_step_graph_tail),
        otherwise by a preparation launch on a side stream.  The following call must pass the same three tensors, unchanged; next_ready: an
        event after which they are complete (e.g. their host-to-device copies), if they are produced on another stream."""
        if self.loss_kind == "CE":
            raise NotImplementedError("graph replay: BCE / BPR only (CE shapes vary with the batch)")
        A = self.arena
        B, S = seq.shape
        if not hasattr(self, "_graphs"):
            self._graphs, self._staged, self._pipe_i = {}, None, 0
        kind = self._sync_kind(grad_hook)
        ktag = "adam" if kind is True else ("grads" if kind is False else "owner")
        pkey = (B, S, self.training) if grad_hook is None else (B, S, self.training, ktag)
        if self._tail_prep_ok() and B <= 8192 and (next_batch is not None or pkey in getattr(self, "_tail_pipes", {})):
            return self._train_step_graph_tail(seq, pos, neg, next_batch, next_ready, grad_hook)
        staged, self._staged = self._staged, None
        hit = staged is not None and staged[0] is seq and staged[1] is pos and staged[2] is neg and staged[5] == (ktag, self.training)
        pipelined = hit or next_batch is not None
        key = (B, S, ktag, self.training) + (((self._pipe_i & 1),) if pipelined else ())
        if hit:
            g = staged[3]
            torch.cuda.current_stream().wait_event(staged[4])
        else:
            if key not in self._graphs:
                self._graphs[key] = self._capture(B, S, with_adam=kind, in_prep=not pipelined)
            g = self._graphs[key]
        if next_batch is not None:
            entry = torch.cuda.Event()
            entry.record()                      # (everything the caller enqueued so far: the next batch exists, the other copy's last replay is done)
        if not hit:
            self._stage(g, seq, pos, neg, A.step + 1)
        g["graph"].replay()
        A.step += 1
        if pipelined:
            self._pipe_i += 1
        if next_batch is not None:
            nseq, npos, nneg = next_batch
            k2 = (nseq.shape[0], nseq.shape[1], ktag, self.training, self._pipe_i & 1)
            if k2 not in self._graphs:
                self._graphs[k2] = self._capture(nseq.shape[0], nseq.shape[1], with_adam=kind, in_prep=False)
            g2 = self._graphs[k2]
            if not hasattr(self, "_prep_stream"):
                self._prep_stream = torch.cuda.Stream()
            ps = self._prep_stream
            ps.wait_event(entry)
            if next_ready is not None:
                ps.wait_event(next_ready)
            with torch.cuda.stream(ps):
                self._stage(g2, nseq, npos, nneg, A.step + 1)
                ev = torch.cuda.Event()
                ev.record(ps)
            self._staged = (nseq, npos, nneg, g2, ev, (ktag, self.training))
        if kind is False:
            self._hook_and_adam(grad_hook)
        self._note_loss(g["loss"], B)
        return g["loss"].squeeze(0)

    # ---- the pipelined form: the NEXT batch is prepared by jobs of THIS step's tail launch (csrc/enc_tail.hip: the preparation depends on the
    #      batch alone, and most of the tail launch's workgroups are done with the item table long before its last one).  Two captured copies
    #      of the step alternate; copy p reads the staging buffers p and its tail launch fills the buffers 1 - p from the addresses the stage
    #      launch in front of it left in a mailbox.  Per step: one stage launch (step scalars, loss fold, weight fragments, mailbox) + one replay.
    def _tail_prep_ok(self):
        return bool(getattr(self, "prep_in_tail", False) and self.fused_item_kernel and self.D == 64 and getattr(self, "fuse_tail", True)
                    and self.encoder == "fused" and self.compact_rows)

    def _tail_pipe(self, B, S, with_adam=True):
        """with_adam (see _sync_kind): False = the step without its optimizer (a gradient hook, then Adam, behind the replay); an OwnerAdam =
        the data-parallel step recorded with it."""
        if not hasattr(self, "_tail_pipes"):
            self._tail_pipes = {}
        key = (B, S, self.training) if with_adam is True else (B, S, self.training, "grads" if with_adam is False else "owner")
        tp = self._tail_pipes.get(key)
        if tp is None:
            nbytes = ops.prep_layout(B, S)[1]
            blobs = [torch.zeros(nbytes, dtype=torch.uint8, device=self.device) for _ in range(2)]
            mail = torch.zeros(ops.MAIL_WORDS, dtype=torch.int64, device=self.device)          # (zero: the captures' warm-up runs prepare nothing)
            graphs = []
            for p in range(2):                                                     # both copies now: a capture's warm-up overwrites its staging buffers
                nxt = ops.next_prep(mail, blobs[1 - p], B, S, max_tiles=self._max_tiles(), split=self._split(), ncu=self._plan_ncu(), tile=self._wave_step(), tile_wgs=self._tile_wgs())
                graphs.append(self._capture(B, S, with_adam=with_adam, in_prep=True, blob=blobs[p], next_prep=nxt))
            tp = self._tail_pipes[key] = dict(blobs=blobs, mail=mail, graphs=graphs, parity=0, staged=None)
        return tp

    def _fresh_pipe(self, tp):
        """A pipe whose staging buffers were last used under another step generation (ParamArena.step): flags and staged batch are dropped."""
        gen = self.arena.step_gen
        if tp.get("gen") != gen:
            if "gen" in tp:
                for b in tp["blobs"]:
                    b.zero_()
                tp["staged"] = None
            tp["gen"] = gen
        return tp

    def _train_step_graph_tail(self, seq, pos, neg, next_batch, next_ready, grad_hook=None):
        A = self.arena
        B, S = seq.shape
        kind = self._sync_kind(grad_hook)
        tp = self._fresh_pipe(self._tail_pipe(B, S, with_adam=kind))
        p = tp["parity"]
        g = tp["graphs"][p]
        st, tp["staged"] = tp["staged"], None
        if not (isinstance(st, tuple) and st[0] is seq and st[1] is pos and st[2] is neg):
            # nobody prepared this batch: the plain preparation launch in front of the step (an epoch's first batch)
            ops.sasrec_batch_prep(seq, pos, neg, blob=tp["blobs"][p], max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(),
                                  ncu=self._plan_ncu())
        if next_batch is not None and tuple(next_batch[0].shape) != (B, S):
            next_batch = None                                                      # (another shape: its own copies' buffers; prepared in front of its step)
        if next_batch is not None and next_ready is not None:
            torch.cuda.current_stream().wait_event(next_ready)
        step = A.step + 1
        sd = (self.seed * 0x9E3779B1 + step * 0x85EBCA77) & 0xFFFFFFFF          # (= _step_seed() once arena.step == step - 1)
        ops.sasrec_step_stage(g["state"], sd, step, self.lr, self.betas[0], self.betas[1], B, S, mail=tp["mail"], next_batch=next_batch,
                              weights=self._prep_weights(B, S), loss_acc=self._take_pending_loss())
        g["graph"].replay()
        A.step += 1
        tp["parity"] = 1 - p
        tp["staged"] = next_batch
        if kind is False:
            self._hook_and_adam(grad_hook)
        self._note_loss(g["loss"], B)
        return g["loss"].squeeze(0)

    def train_step_graph_sampled(self, ticket, next_ticket=None):
        """The captured step on a batch the preparation launch SAMPLES itself (recboard_amd.sampler.DeviceSeqSampler(fused=True) hands out
        tickets instead of tensors): one sample + prepare launch, one graph replay -- no sampler launch, no batch tensors.
        next_ticket (the FOLLOWING call's ticket, if the caller has it): that batch is sampled and prepared by jobs of this step's tail launch
        (the pipelined form of train_step_graph: _tail_pipe); the following call must pass the same ticket object."""
        A = self.arena
        B, S = ticket.B, ticket.S
        if self._tail_prep_ok() and B <= 8192 and (next_ticket is not None or (B, S, self.training) in getattr(self, "_tail_pipes", {})):
            tp = self._fresh_pipe(self._tail_pipe(B, S))
            p = tp["parity"]
            g = tp["graphs"][p]
            st, tp["staged"] = tp["staged"], None
            if st is not ticket:
                ops.sasrec_sample_prep(ticket.inter, ticket.order, ticket.b0, B, S, ticket.seed, ticket.step, tp["blobs"][p], max_tiles=self._max_tiles(),
                                       split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(), ncu=self._plan_ncu(), users=ticket.users)
            if next_ticket is not None and (next_ticket.B, next_ticket.S) != (B, S):
                next_ticket = None
            ops.sasrec_step_stage(g["state"], self._step_seed(), A.step + 1, self.lr, self.betas[0], self.betas[1], B, S, mail=tp["mail"],
                                  next_ticket=next_ticket, weights=self._prep_weights(B, S), loss_acc=self._take_pending_loss())
            g["graph"].replay()
            A.step += 1
            tp["parity"] = 1 - p
            tp["staged"] = next_ticket
            self._note_loss(g["loss"], B)
            return g["loss"].squeeze(0)
        key = (B, S, True, self.training)
        if not hasattr(self, "_graphs"):
            self._graphs, self._staged, self._pipe_i = {}, None, 0
        if key not in self._graphs:
            self._graphs[key] = self._capture(B, S, with_adam=True)
        g = self._graphs[key]
        ops.sasrec_sample_prep(ticket.inter, ticket.order, ticket.b0, B, S, ticket.seed, ticket.step, g["blob"], state=g["state"],
                               seed=self._step_seed(), step=A.step + 1, lr=self.lr, beta1=self.betas[0], beta2=self.betas[1],
                               max_tiles=self._max_tiles(), split=self._split(), tile=self._wave_step(), tile_wgs=self._tile_wgs(), ncu=self._plan_ncu(),
                               weights=self._prep_weights(B, S) if g["in_prep"] else None, users=ticket.users, loss_acc=self._take_pending_loss())
        g["graph"].replay()
        A.step += 1
        self._note_loss(g["loss"], B)
        return g["loss"].squeeze(0)

    # ---- CoachForSASRec.train_per_epoch body (SASRec/main.py:243-250): zero_grad, backward, Adam step
    def train_step(self, seq, pos, neg, aux=None, grad_hook=None):
        return self.train_step_fused(seq, pos, neg, aux, grad_hook)
