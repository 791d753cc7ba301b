"""DeepFM on the engine: host-side mirror of the reference's `DeepFM` (DeepFM/main.py:127-219) and of
`CoachForDeepFM.train_per_epoch`'s step body (:256-276).  Every heavy op of the step is a librecengine kernel (what is
left to torch: adding two [B] vectors, the gradient-norm reduction of the clip, the eval-time sigmoid):

  multi-field embedding bag + FM second-order term + logistic regression   re_fm_bag_fwd / _bwd          (DeepFM/main.py:58-62,80-85,204-206)
  MLP  Linear -> BatchNorm1d -> ReLU -> Dropout (x3) -> Linear(., 1)       re_gemm_f32 + re_bn_relu_drop_*  (:103-124,151-164)
  BCELoss4Logits(mean) and its gradient                                     re_bce_logits                 (:214)
  table gradients                                                           re_scatter_add_rows (one launch per table, all fields)
  clip_grad_norm_(.., 10) + Adam with the two weight-decay groups           re_rows_sqnorm-style norm + re_adam_step x2   (:187-199,267-268)

Layout: ONE parameter arena = [ T (sum count_f x D) | TL (sum count_f) | pad ][ MLP weights, biases, BN affine | LR bias ];
the first segment is the reference's "embeddings" optimizer group (weight_decay = embedding_decay), the second its
"other" group -- two Adam launches.  The F per-field tables are one concatenated table: `field f, id i` is row
`offsets[f] + i`; `tables()` / `tables_lr()` give the reference's per-field views.
"""
import math
from collections import OrderedDict

import torch

from . import ops


class DeepFMEngine:
    def __init__(self, counts, embedding_dim=10, hidden_dims=(400, 400, 400), batch_norm=True, hidden_dropout_rate=0.0,
                 lr=1e-3, embedding_decay=0.05, weight_decay=0.0, betas=(0.9, 0.999), device="cuda", seed=1):
        self.counts, self.F, self.D = list(counts), len(counts), embedding_dim
        self.device = torch.device(device)
        self.lr, self.emb_decay, self.wd, self.betas = lr, embedding_decay, weight_decay, betas
        self.bn, self.p_drop, self.seed = batch_norm, hidden_dropout_rate, seed
        off = [0]
        for c in self.counts[:-1]:
            off.append(off[-1] + c)
        self.rows = sum(self.counts)
        self.offsets = torch.tensor(off, dtype=torch.int64, device=self.device)
        dims = [self.F * self.D] + list(hidden_dims)
        self.dims = dims
        shapes = OrderedDict()
        shapes["T"] = (self.rows, self.D)
        shapes["TL"] = (self.rows, 1)
        self.n_emb = sum((math.prod(s) + 3) // 4 * 4 for s in shapes.values())          # embeddings group ends here
        for i, (a, b) in enumerate(zip(dims[:-1], dims[1:])):
            shapes[f"dnn.{i}.linear.weight"] = (b, a)
            shapes[f"dnn.{i}.linear.bias"] = (b,)
            if batch_norm:
                shapes[f"dnn.{i}.bn.weight"] = (b,)
                shapes[f"dnn.{i}.bn.bias"] = (b,)
        nl = len(dims) - 1
        shapes[f"dnn.{nl}.weight"] = (1, dims[-1])
        shapes[f"dnn.{nl}.bias"] = (1,)
        shapes["fm.lr_layer.bias"] = (1,)
        self.shapes, self.nl = shapes, nl
        self.off = OrderedDict()
        o = 0
        for k, shp in shapes.items():
            self.off[k] = o
            o += (math.prod(shp) + 3) // 4 * 4
        self.numel = o
        self.data = torch.zeros(o, device=self.device)
        self.grad, self.m, self.v = (torch.zeros_like(self.data) for _ in range(3))
        self.P = self._views(self.data)
        self.G = self._views(self.grad)
        self.T, self.TL, self.bias = self.P["T"], self.P["TL"], self.P["fm.lr_layer.bias"]
        self.gT, self.gTL, self.gbias = self.G["T"], self.G["TL"], self.G["fm.lr_layer.bias"]
        self.running = {}
        if batch_norm:
            for i, b in enumerate(dims[1:]):
                self.running[i] = (torch.zeros(b, device=self.device), torch.ones(b, device=self.device))
        g = torch.Generator().manual_seed(seed)
        with torch.no_grad():   # DeepFM.reset_parameters (DeepFM/main.py:172-182)
            for k, p in self.P.items():
                if k in ("T", "TL"):
                    p.copy_((torch.randn(p.shape, generator=g) * 1e-4).to(self.device))
                elif k.endswith("linear.weight") or k == f"dnn.{nl}.weight":
                    std = math.sqrt(2.0 / (p.shape[0] + p.shape[1]))
                    p.copy_((torch.randn(p.shape, generator=g) * std).to(self.device))
                elif k.endswith("bn.weight"):
                    p.fill_(1.0)
        self.step = 0
        self.training = True

    def _views(self, buf):
        return OrderedDict((k, buf[o:o + math.prod(self.shapes[k])].view(self.shapes[k])) for k, o in self.off.items())

    # ---- per-field views (reference state-dict granularity) and state-dict loading
    def tables(self):
        return [self.T[o:o + c] for o, c in zip(self.offsets.tolist(), self.counts)]

    def tables_lr(self):
        return [self.TL[o:o + c] for o, c in zip(self.offsets.tolist(), self.counts)]

    def load_dnn_state_dict(self, sd):
        """`dnn.*` entries of the reference state_dict (incl. BatchNorm running statistics)."""
        for k, p in self.P.items():
            if k.startswith("dnn."):
                p.copy_(torch.as_tensor(sd[k[4:]]).to(self.device).view(p.shape))
        for i in self.running:
            self.running[i][0].copy_(torch.as_tensor(sd[f"{i}.bn.running_mean"]).to(self.device))
            self.running[i][1].copy_(torch.as_tensor(sd[f"{i}.bn.running_var"]).to(self.device))

    def state_dict(self):
        """The reference's state-dict granularity: per-field tables, the LR bias, the MLP incl. BatchNorm running statistics."""
        sd = OrderedDict()
        for f, (t, tl) in enumerate(zip(self.tables(), self.tables_lr())):
            sd[f"fields.{f}.embeddings.weight"] = t.detach().clone()
            sd[f"fields.{f}.embeddings_lr.weight"] = tl.detach().clone()
        sd["fm.lr_layer.bias"] = self.bias.detach().clone()
        for k, p in self.P.items():
            if k.startswith("dnn."):
                sd[k] = p.detach().clone()
        for i, (rm, rv) in self.running.items():
            sd[f"dnn.{i}.bn.running_mean"], sd[f"dnn.{i}.bn.running_var"] = rm.clone(), rv.clone()
        return sd

    def load_state_dict(self, sd):
        with torch.no_grad():
            for f, (t, tl) in enumerate(zip(self.tables(), self.tables_lr())):
                t.copy_(torch.as_tensor(sd[f"fields.{f}.embeddings.weight"]).to(self.device))
                tl.copy_(torch.as_tensor(sd[f"fields.{f}.embeddings_lr.weight"]).to(self.device))
            self.bias.copy_(torch.as_tensor(sd["fm.lr_layer.bias"]).to(self.device))
            self.load_dnn_state_dict({k[4:]: v for k, v in sd.items() if k.startswith("dnn.")})

    def adam_state_dict(self):
        return {"m": self.m.clone(), "v": self.v.clone(), "step": self.step, "lr": self.lr}

    def load_adam_state_dict(self, st):
        self.m.copy_(st["m"]); self.v.copy_(st["v"]); self.step = int(st["step"]); self.lr = float(st.get("lr", self.lr))

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def _step_seed(self):
        return (self.seed * 0x9E3779B1 + (self.step + 1) * 0x85EBCA77) & 0xFFFFFFFF

    # ---- forward: keeps what the backward needs in `tape`
    def encode(self, x, seed_dev=None, labels=None, _defer_final=False):
        """-> (logits [B], tape).  DeepFM/main.py:201-209.  seed_dev: the dropout seed as a device word (captured steps).
        labels (training): the criterion runs in the last layer's launch; tape["loss" / "dlogit" / "dsum"] hold its results."""
        P = self.P
        # (what the backward's table gradient wants of the batch: the ids field-major for re_fm_table_grad, or the destination rows for the
        #  general sorted scatter-add where the shapes are beyond it)
        rows = keys_t = None
        if labels is not None:
            if ops.fm_table_grad_ok(x.shape[0], self.F, self.D, max(self.counts)):
                keys_t = torch.empty(x.numel(), dtype=torch.int32, device=x.device)
            else:
                rows = torch.empty(x.numel(), dtype=torch.int64, device=x.device)
        E, fm_lr = ops.fm_bag_fwd(self.T, self.TL.reshape(-1), self.bias, self.offsets, x, rows_out=rows, keys_t=keys_t)
        tape = {"E": E, "layers": [], "rows": rows, "keys_t": keys_t}
        h, sd = E, self._step_seed()
        for i in range(self.nl):
            # bn(linear(x)) in training mode: the batch statistics' per-64-row partials come out of the GEMM's epilogue (re_gemm_f32_colstats)
            fused = ops.gemm_colstats(h, P[f"dnn.{i}.linear.weight"], True, P[f"dnn.{i}.linear.bias"]) if (self.bn and self.training) else None
            z, cs = fused if fused is not None else (ops.gemm(h, P[f"dnn.{i}.linear.weight"], transB=True, bias=P[f"dnn.{i}.linear.bias"]), None)
            rm, rv = self.running[i] if self.bn else (None, None)
            a, stats = ops.bn_relu_drop_fwd(z, P.get(f"dnn.{i}.bn.weight"), P.get(f"dnn.{i}.bn.bias"), rm, rv, self.training,
                                            self.p_drop, sd, stream_id=100 + i, seed_dev=seed_dev, colstats=cs)
            tape["layers"].append((h, z, a, stats))
            h = a
        tape["h_last"], tape["fm_lr"] = h, fm_lr
        # logits = lr + fm + dnn: the last Linear(., 1) as a row dot with the FM / LR term added in the same launch (re_mlp_head_fwd)
        w_last = P[f"dnn.{self.nl}.weight"].reshape(-1)
        if labels is None:
            return ops.mlp_head_fwd(h, w_last, P[f"dnn.{self.nl}.bias"], fm_lr), tape
        # (_defer_final: the criterion's sums -- loss, the two biases' gradient -- are finished by the head's gated backward launch, which
        #  forward_backward issues next: tape["final"] is what it needs for that)
        r = ops.mlp_head_fwd(h, w_last, P[f"dnn.{self.nl}.bias"], fm_lr, labels.reshape(-1).to(torch.float32).contiguous(),
                             dsum=self.G[f"dnn.{self.nl}.bias"], dsum2=self.gbias, defer_final=_defer_final)
        logits, tape["loss"], tape["dlogit"], tape["dsum"] = r[:4]
        tape["final"] = r[4] if _defer_final else None
        return logits, tape

    def recommend_from_pool(self, x):
        """sigmoid(logits) [B, 1]  (DeepFM/main.py:217-219)."""
        return torch.sigmoid(self.encode(x)[0]).unsqueeze(1)

    def forward_backward(self, x, labels, seed_dev=None, _tables_zeroed=False):
        """loss + every gradient into the gradient arena.  DeepFM/main.py:211-215, 264-266.
        _tables_zeroed: the two table gradients were cleared by the caller (the captured step: by the launch that brings the batch in)."""
        P, G = self.P, self.G
        # last layer + criterion fused (re_mlp_head_fwd with labels: loss, dlogit and its sum -- the last bias's gradient); then dW = dl^T h and
        # da = dl w in one pass over h (re_mlp_head_bwd)
        nl = self.nl
        logits, tape = self.encode(x, seed_dev, labels=labels, _defer_final=self.bn and self.training and nl > 0)
        loss, dlogit, dsum = tape["loss"], tape["dlogit"], tape["dsum"]
        p_drop = self.p_drop if self.training else 0.0
        w_last, gw_last = P[f"dnn.{nl}.weight"].reshape(-1), G[f"dnn.{nl}.weight"].reshape(-1)
        # BatchNorm in training mode: the gate g = dropout'(relu'(.)) x incoming gradient and its column sums (sum g, sum g xhat) come out of the
        # launch that PRODUCES the incoming gradient (the head's backward for the last block, the dx product's epilogue for the others), and one
        # pass (re_bn_bwd_apply) finishes dgamma / dbeta and turns g into dz.  Otherwise: da, then re_bn_relu_drop_bwd's three launches.
        fused = self.bn and self.training and nl > 0
        da = g = part = None
        pending = []
        if fused:
            g, part = ops.mlp_head_bwd_gated(dlogit, tape["h_last"], w_last, tape["layers"][nl - 1][1], tape["layers"][nl - 1][3], p_drop,
                                             final=tape["final"])
        else:
            da = ops.mlp_head_bwd(dlogit, tape["h_last"], w_last, gw_last)
        for i in reversed(range(nl)):
            h, z, a, stats = tape["layers"][i]
            if g is not None:
                dz = ops.bn_bwd_apply(g, z, P[f"dnn.{i}.bn.weight"], stats, part, G[f"dnn.{i}.bn.weight"], G[f"dnn.{i}.bn.bias"],
                                      extra_out=gw_last if part.shape[1] == 3 else None)
            else:
                dz, _, _ = ops.bn_relu_drop_bwd(da, a, z, P.get(f"dnn.{i}.bn.weight"), stats, p_drop,
                                                dgamma=G.get(f"dnn.{i}.bn.weight"), dbeta=G[f"dnn.{i}.bn.bias"] if self.bn else G[f"dnn.{i}.linear.bias"])
            ops.gemm(dz, h, transA=True, out=G[f"dnn.{i}.linear.weight"], defer=pending)   # dW = dz^T x (split-K: reduced with the others below)
            # (the Linear bias in front of a BatchNorm: its gradient sum_m dz[m, :] is zero in exact arithmetic -- BatchNorm's backward removes the
            #  column mean -- and pure cancellation noise (~1e-10 of the model's gradient scale) as autograd computes it; it stays exactly zero
            #  here: the arena is zero-initialised and nothing writes these entries)
            W = P[f"dnn.{i}.linear.weight"]
            r = None
            if fused and i > 0:                                                        # dx = dz W, gated for the block underneath
                hp, zp, ap, sp = tape["layers"][i - 1]
                r = ops.gemm_gated(dz, W, ap, zp, sp, 1.0 / (1.0 - p_drop) if p_drop > 0 else 1.0)
            if r is not None:
                g, part = r
            else:
                g, da = None, ops.gemm(dz, W)
        ops.gemm_reduce_many(pending)        # the weight-gradient products' split-K partials: one reduction launch for all layers
        gE, gL = ops.fm_bag_bwd(tape["E"], da, dlogit, self.F, self.D)
        if tape["keys_t"] is not None:
            # a field's B keys live in the field's own row range: a workgroup per slice of rows collects its keys and sums them per row
            # (re_fm_table_grad: one launch behind one zero fill, instead of the general sorted scatter-add's twelve)
            if not _tables_zeroed:
                self.grad[:self.n_emb].zero_()
            B = tape["keys_t"].numel() // self.F
            sl = self.__dict__.setdefault("_slices", {})
            if B not in sl:
                sl[B] = ops.fm_table_slices(self.counts, B, self.device)
            ops.fm_table_grad(tape["keys_t"], self.offsets, sl[B], gE, gL, self.gT, self.gTL.reshape(-1))
            return loss.squeeze(0)
        rows = tape["rows"]
        # both table gradients follow the same destination rows: ONE sort (re_scatter_plan), two segmented sums (re_scatter_apply)
        ws = ops.scatter_workspace(rows.numel(), self.D, self.rows, rows.device)
        ops.scatter_plan(rows, self.D, self.rows, ws)
        ops.scatter_apply(gE.reshape(-1, self.D), self.rows, self.gT, ws, accumulate=False)
        ops.scatter_apply(gL.reshape(-1, 1), self.rows, self.gTL, ws, accumulate=False)
        return loss.squeeze(0)

    def train_step(self, x, labels, max_norm=10.0, _state=None):
        """forward, backward, clip_grad_norm_(.., 10), Adam with the reference's two groups (DeepFM/main.py:187-199, 264-268)."""
        loss = self.forward_backward(x, labels, seed_dev=_state, _tables_zeroed=_state is not None and self._stage_zeroes(x))
        # clip_grad_norm_(.., max_norm) + Adam with the reference's two weight-decay groups: the square norm's partials, then ONE launch whose
        # workgroups form the coefficient from them and apply it on their way through the gradient (which they leave clipped in the arena, as
        # p.grad is after the reference's step)
        out = self.__dict__.setdefault("_clip", torch.empty(2, dtype=torch.float32, device=self.device))
        b1, b2 = self.betas
        if _state is not None:       # captured: seed and Adam scalars are device words; the host counts the step (train_step_graph)
            ops.adam_step_clip2(self.data, self.grad, self.m, self.v, self.n_emb, max_norm, hyper=_state.view(torch.float32)[2:4], beta1=b1, beta2=b2,
                                wd_first=self.emb_decay, wd_rest=self.wd, out=out)
            return loss
        self.step += 1
        ops.adam_step_clip2(self.data, self.grad, self.m, self.v, self.n_emb, max_norm, step=self.step, lr=self.lr, beta1=b1, beta2=b2,
                            wd_first=self.emb_decay, wd_rest=self.wd, out=out)
        return loss

    def _stage_zeroes(self, x):
        """The captured step's table gradients are cleared in front of the replay (capture.CapturedStep: zeros) when re_fm_table_grad is the
        kernel that fills them (it writes only the rows the batch points at)."""
        return ops.fm_table_grad_ok(x.shape[0], self.F, self.D, max(self.counts))

    def train_step_graph(self, x, labels, max_norm=10.0):
        """train_step as one hipGraph replay (recboard_amd/capture.py): two copies into the static batch, one launch for the step's seed and
        Adam scalars, one replay.  Same results as the eager step (same seeds, same launches)."""
        from .capture import captured
        labels = labels.reshape(-1)
        restore = [self.data, self.m, self.v] + [t for pair in self.running.values() for t in pair]
        x = x.contiguous()
        zeros = [self.grad[:self.n_emb]] if self._stage_zeroes(x) else []
        # (the static label buffer is fp32; int64 labels are converted on their way in by the staging launch)
        g = captured(self, ("train", float(max_norm)), lambda xx, yy, st: self.train_step(xx, yy, max_norm, _state=st), (x, labels), restore, zeros,
                     static_dtypes=(x.dtype, torch.float32))
        sd = self._step_seed()
        self.step += 1
        return g((x, labels), sd, self.step, self.lr, self.betas[0], self.betas[1])
