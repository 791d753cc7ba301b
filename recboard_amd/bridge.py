"""Engine routing for the `freerec`-compatible surface (freerec/launcher.py): recognise a model the engine has a fused step for, PROVE on one
batch that the engine's step is the step the script itself would take, and only then drive the model through the engine with the MODULE's
parameters living in the engine's arena.

Models (the four north-star scripts):
  SASRec    SASRec/main.py:53-228      `Item.embeddings` [N + 1, D], `Position`, `attnLNs / attnLayers / fwdLNs / fwdLayers`, `lastLN`, BCE / BPR / CE
  MF-BPR    MF-BPR/main.py:25-131      `User.embeddings`, `Item.embeddings`, BPRLoss; nothing else trainable
  LightGCN  LightGCN/main.py:27-172    the same + a sparse `Adj` buffer + `num_layers`; loss = rec + cfg.weight_decay * emb, optimizer without decay
  DeepFM    DeepFM/main.py:127-276     per-field `embeddings` / `embeddings_lr`, `fm.lr_layer.bias`, `dnn` (MLPBlock x n + Linear), BCE; two decay groups, clip 10

ADOPTION IS PROBED, NOT ASSUMED.  Attribute names say what a module calls its parts, not what it computes with them (a SASRec variant with
a key-padding mask has the same names), and a Coach subclass may build another optimizer, clip, accumulate or schedule.  So before an
adapter is returned:
  1. the optimizer the script built must be a plain torch.optim.Adam (no amsgrad / maximize, eps 1e-8) over exactly the model's parameters,
     with one lr and one beta pair; the engine is given ITS numbers (lr, betas, per-group weight decay), not cfg's;
  2. the script's own `train_per_epoch` runs on two copies of the first training batch with dropout switched off, on the module's torch
     code, while the optimizer / scheduler calls are recorded; the gradients it hands to `optimizer.step()` (after any clipping, with any
     extra loss terms) are kept; module, optimizer, scheduler and monitors are restored afterwards;
  3. the engine takes one step on the same batch from the same parameters: its gradients must equal the script's own step in DOUBLE
     precision, tensor by tensor, to GRAD_TOL = 2e-2 in the L2 sense and 10 x that (0.2 of the tensor's largest entry) for the worst one
     -- or 4 x the script's own fp32 rounding of that tensor, or 1e-6 of the model-wide largest gradient, where those are larger -- and its
     parameter update must equal torch's Adam formula on those gradients with the optimizer's own numbers.  The bound is wide because a
     ReLU pre-activation within rounding of zero may fall on the other side in the engine's arithmetic (one flipped gate moves a whole row
     of the FFN's first map); measured on the four reference scripts the distance is 7e-6 (median) to 3e-3 (worst).  What the bound does
     NOT catch: a script whose step differs from the engine's by a small extra term (an auxiliary loss scaled by <= 1e-2, light label
     smoothing) is accepted and then trained WITHOUT that term -- `--engine module` is the answer for such scripts; the probe is one
     step deep, so a non-default beta pair is read from the optimizer object, not observed;
  4. the recorded call pattern must be one the adapter replays: one `optimizer.step()` per batch; the scheduler not at all, or once per epoch
     (in front of the loop with `coach._best`, as DeepFM/main.py:256, or behind it without arguments).
Anything else -- and any exception on the way -- logs a warning that names the model and the reason, and the script runs on its own torch
code.  Under torchrun (`ddp.is_distributed()`) the engines are not attached: their captured steps carry no gradient all-reduce.

After adoption every module parameter IS the arena view of its name (`p.data = view`): the script's own `encode / recommend_from_full`,
`state_dict()` and checkpoints read what the engine trains; the learning rate of an epoch is read from `coach.optimizer.param_groups`."""
import copy
import math
import os
import warnings

import torch
import torch.nn as nn

from .deepfm import DeepFMEngine
from .gen import LightGCNEngine, MFEngine
from .sasrec import SASRecEngine

GRAD_TOL = 2e-2        # engine gradient vs the script's own step in double precision, per tensor in the L2 sense (worst entry: 10 x).  Measured
                       # (scripts/probe_margin.py, B = 512, twelve random initialisations of the genuine SASRec script: 336 tensors): median 7e-6,
                       # largest 6.2e-3 (the FFN's first map: a relu unit within rounding of zero falls on the other side in one of the two
                       # arithmetics and moves its whole row), worst entry 3e-2 of the tensor's largest; a look-alike with other arithmetic
                       # (layer-normed keys): median 0.26, up to 1.0.  1e-3 refused the genuine script in two fresh processes of ten.
GRAD_TOL_F32 = 2e-2    # ... vs the script's fp32 step when the module does not run in double (ROCm aten's fp32 GEMMs alone are off by up to 7e-3)
# Per-tensor classes (round 6; the same measurement, gpurun_out/probe_margin.txt, by tensor name).  What moves a genuine script's gradient away
# from its fp64 self is a ReLU gate that falls on the other side of zero in one of the two arithmetics; the flip moves the unit's own row of the
# map in FRONT of the ReLU by its whole contribution and everything upstream of it by that token's share:
#   the map whose output the ReLU gates (SASRec fwdLayers.l.conv1.*, DeepFM dnn.i.linear.weight): largest 6.2e-3 -> GRAD_TOL = 2e-2
#   tensors no gate can move (behind the last ReLU, or models without one: SASRec lastLN.*, MF-BPR / LightGCN tables, DeepFM's LR terms and
#     last layer): largest 2.1e-6 -> GRAD_TOL_SMOOTH = 1e-4 (a look-alike with layer-normed keys is 6.4e-2 .. 0.21 off THERE)
#   everything else (embeddings, attention maps, LayerNorms upstream of an FFN): largest 2.7e-3 -> GRAD_TOL_UPSTREAM = 1e-2 (look-alike: >= 0.14)
GRAD_TOL_SMOOTH = 1e-4
GRAD_TOL_UPSTREAM = 1e-2
UPDATE_TOL = 2e-4      # engine update vs torch's Adam formula on the engine's gradient, relative to lr


class Refused(Exception):
    """The model is left to its own torch code; the message says why."""


def _log(msg):
    try:
        from freerec import utils
        utils.infoLogger(msg)
    except Exception:  # noqa: BLE001
        print(msg)


# ---------------------------------------------------------------------------------------------------------------------------------------
# what the script's optimizer says
# ---------------------------------------------------------------------------------------------------------------------------------------
class OptSpec:
    """lr, betas and the weight decay of every parameter, read from the torch optimizer the script built."""

    def __init__(self, coach, module):
        opt = getattr(coach, "optimizer", None)
        if type(opt) is not torch.optim.Adam:
            raise Refused(f"optimizer is {type(opt).__name__}, the engines step with torch.optim.Adam's rule only")
        if len(opt.state):
            raise Refused("the optimizer already holds state")
        params = {id(p): n for n, p in module.named_parameters()}
        seen = set()
        self.wd = {}
        lrs, betas = set(), set()
        for g in opt.param_groups:
            if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or abs(g.get("eps", 1e-8) - 1e-8) > 0:
                raise Refused("Adam with amsgrad / maximize / eps != 1e-8")
            if g.get("decoupled_weight_decay"):
                raise Refused("decoupled weight decay")
            lrs.add(float(g["lr"])); betas.add(tuple(float(b) for b in g["betas"]))
            for p in g["params"]:
                if id(p) not in params:
                    raise Refused("the optimizer steps a tensor that is not a parameter of the model")
                seen.add(id(p))
                self.wd[params[id(p)]] = float(g["weight_decay"])
        missing = [n for i, n in params.items() if i not in seen and dict(module.named_parameters())[n].requires_grad]
        if missing:
            raise Refused(f"parameters outside the optimizer: {missing[:3]}")
        if len(lrs) != 1 or len(betas) != 1:
            raise Refused("parameter groups with different lr / betas")
        self.lr, self.betas = lrs.pop(), betas.pop()

    def one_wd(self, names):
        w = {self.wd[n] for n in names}
        if len(w) != 1:
            raise Refused(f"mixed weight decay inside one engine group: {sorted(w)}")
        return w.pop()


def _pool_tensor(pool, device):
    """An evaluation batch's candidate pools (`ranking="pool"`: a [B, P] tensor, or equal-length lists) as int64 [B, P] on the device."""
    t = pool if torch.is_tensor(pool) else torch.as_tensor(pool, dtype=torch.int64)
    return t.to(device=device, dtype=torch.int64).reshape(t.shape[0], -1).contiguous()


def _current_lr(coach, fallback):
    try:
        return float(coach.optimizer.param_groups[0]["lr"])
    except Exception:  # noqa: BLE001
        return fallback


# ---------------------------------------------------------------------------------------------------------------------------------------
# adapters
# ---------------------------------------------------------------------------------------------------------------------------------------
class _Adapter:
    """Common part: the name -> (parameter view, gradient view) maps, binding, the probe's engine step, the scheduler replay."""
    kind = "?"
    sched_mode = None          # None | "front_best" | "back"

    def named_views(self):
        raise NotImplementedError

    def named_grads(self):
        raise NotImplementedError

    def grad_class(self, name):
        """Which of the probe's three bounds a parameter's gradient is held to (GRAD_TOL*): "gate" -- the map whose output a ReLU gates;
        "smooth" -- no ReLU gate can move it; "upstream" -- everything else."""
        return "gate"

    def load_from_module(self):
        with torch.no_grad():
            named = dict(self.module.named_parameters())
            for k, v in self.named_views().items():
                v.copy_(named[k].detach().reshape(v.shape))

    def bind(self):
        named = dict(self.module.named_parameters())
        for k, v in self.named_views().items():
            named[k].data = v.detach().view(named[k].shape)

    def begin_epoch(self, coach):
        if self.sched_mode == "front_best":
            coach.lr_scheduler.step(coach._best)
        self._set_lr(_current_lr(coach, self.spec.lr))

    def end_epoch(self, coach):
        if self.sched_mode == "back":
            coach.lr_scheduler.step()

    def reset_ranking_buffers(self):
        self.eng.eval() if hasattr(self.eng, "eval") else None
        self.eng.reset_ranking_buffers()

    def _set_lr(self, lr):
        self.eng.lr = lr

    # ---- optimizer state in torch.optim.Adam's state_dict shape over the SCRIPT's parameter groups: what `checkpoint.tar` holds under
    #      "optimizer" whichever way the model is trained, so a checkpoint moves between `--engine auto`, `--engine module` and the reference
    def named_moments(self):
        """-> (name -> first-moment view, name -> second-moment view) under the module's parameter names."""
        A = self.eng.arena
        return A.views(A.m), A.views(A.v)

    def _get_step(self):
        return int(self.eng.arena.step)

    def _set_step(self, n):
        self.eng.arena.step = int(n)

    def optimizer_state(self, coach):
        ids = {id(p): n for n, p in self.module.named_parameters()}
        m, v = self.named_moments()
        step = float(self._get_step())
        state, groups, i = {}, [], 0
        for g in coach.optimizer.param_groups:
            gi = {k: val for k, val in g.items() if k != "params"}
            gi["lr"] = float(getattr(self.eng, "lr", g["lr"]))         # (the engine's current rate: a plateau schedule may have lowered it)
            gi["params"] = []
            for p in g["params"]:
                n = ids[id(p)]
                if step > 0:
                    state[i] = {"step": torch.tensor(step), "exp_avg": m[n].detach().clone().reshape(p.shape), "exp_avg_sq": v[n].detach().clone().reshape(p.shape)}
                gi["params"].append(i)
                i += 1
            groups.append(gi)
        return {"state": state, "param_groups": groups}

    def load_optimizer_state(self, coach, sd):
        if "param_groups" not in sd:                                   # (round-4 DeepFM checkpoints: {m, v, step, lr} arena copies)
            self.eng.load_adam_state_dict(sd)
            lr = float(sd.get("lr", self.eng.lr))
        else:
            ids = {id(p): n for n, p in self.module.named_parameters()}
            m, v = self.named_moments()
            step, i = 0, 0
            with torch.no_grad():
                for g in coach.optimizer.param_groups:
                    for p in g["params"]:
                        st = sd["state"].get(i)
                        n = ids[id(p)]
                        if st is not None:
                            m[n].copy_(st["exp_avg"].to(m[n].device).reshape(m[n].shape))
                            v[n].copy_(st["exp_avg_sq"].to(v[n].device).reshape(v[n].shape))
                            step = int(st["step"])
                        else:
                            m[n].zero_(); v[n].zero_()
                        i += 1
            self._set_step(step)
            lr = float(sd["param_groups"][0]["lr"])
        # the script's optimizer is where every epoch's learning rate is read from (begin_epoch): the saved rate goes there too
        for g, gs in zip(coach.optimizer.param_groups, sd.get("param_groups") or [{}] * len(coach.optimizer.param_groups)):
            g["lr"] = float(gs.get("lr", lr))
        self._set_lr(lr)


class SASRecAdapter(_Adapter):
    kind = "SASRec"

    @staticmethod
    def plan(coach, module, spec):
        """Structure and optimizer checks (no GPU work) -> the engine's constructor arguments."""
        need = ("Item", "Position", "attnLNs", "attnLayers", "fwdLNs", "fwdLayers", "lastLN", "criterion")
        if not all(hasattr(module, a) for a in need):
            raise Refused("not SASRec-shaped")
        loss = {"BCELoss4Logits": "BCE", "BPRLoss": "BPR", "CrossEntropy4Logits": "CE"}.get(type(module.criterion).__name__)
        E = module.Item.embeddings.weight
        D, S, L = E.shape[1], module.Position.weight.shape[0], len(module.attnLayers)
        mha = module.attnLayers[0]
        if loss is None or getattr(module.criterion, "reduction", "mean") != "mean":
            raise Refused(f"criterion {type(module.criterion).__name__} / reduction")
        if D not in (64, 128) or S > 64 or L > 4 or mha.num_heads != 1 or not hasattr(module.fwdLayers[0], "conv1") or E.shape[0] != module.Item.count + 1:
            raise Refused(f"shape outside the fused encoder kernels (D={D}, maxlen={S}, blocks={L}, heads={mha.num_heads})")
        cfg = coach.cfg
        p = float(module.embdDropout.p) if hasattr(module, "embdDropout") else float(cfg.get("dropout_rate", 0.0))
        names = [k for k, _ in module.named_parameters()]
        from .sasrec import param_shapes
        want = list(param_shapes(module.Item.count, S, D, L))
        extra, missing = [k for k in names if k not in want], [k for k in want if k not in names]
        if extra or missing:
            raise Refused(f"parameter set differs from SASRec's: extra {extra[:3]}, missing {missing[:3]}")
        return dict(num_items=module.Item.count, maxlen=S, embedding_dim=D, num_blocks=L, dropout_rate=p, loss=loss, lr=spec.lr,
                    weight_decay=spec.one_wd(names), betas=spec.betas, seed=int(cfg.get("seed", 1)))

    def __init__(self, coach, module, spec):
        kw = self.plan(coach, module, spec)
        self.module, self.spec, self.loss = module, spec, kw["loss"]
        self.eng = SASRecEngine(device=coach.device, **kw)
        self.load_from_module()

    def named_views(self):
        return self.eng.params

    def named_grads(self):
        return self.eng.arena.views(self.eng.arena.grad)

    def _batch(self, coach, data):
        dev = coach.device
        seq = data[coach.ISeq].to(dev)
        return seq, data[coach.IPos].to(dev).reshape(seq.shape), data[coach.INeg].to(dev).reshape(seq.shape)

    def grad_class(self, name):
        return "gate" if ".conv1." in name else ("smooth" if name.startswith("lastLN") else "upstream")

    def probe_step(self, coach, data):
        eng, A = self.eng.train(), self.eng.arena
        p, eng.p_drop = eng.p_drop, 0.0
        try:
            eng.train_step(*self._batch(coach, data))
        finally:
            eng.p_drop = p

    def reset_state(self):
        A = self.eng.arena
        A.m.zero_(); A.v.zero_(); A.step = 0
        for a in ("_graphs", "_tail_pipes", "_staged"):
            if hasattr(self.eng, a):
                delattr(self.eng, a)

    def wants_fused_sampler(self):
        return self.loss != "CE" and hasattr(self.eng, "train_step_graph_sampled")

    def train_epoch(self, coach, epoch):
        from .coach import _lookahead
        self.begin_epoch(coach)
        eng = self.eng.train()
        n = 0
        if self.loss == "CE":           # CE over the catalog: the step's shapes follow the batch's number of real positions -- eager launches
            tot = torch.zeros((), device=coach.device)
            for data in coach.dataloader:
                seq, pos, neg = self._batch(coach, data)
                tot.add_(eng.train_step(seq, pos, neg).reshape(()), alpha=seq.shape[0])
                n += seq.shape[0]
        else:
            eng.begin_loss_accumulation()   # (every step's loss x its batch size is folded into one device word by the next step's stage launch)

            def batches():
                for data in coach.dataloader:
                    if "Sample" in data:      # the device sampler's ticket (freerec pipe -> .to_(device)): a step's launches sample the batch
                        yield data["Sample"]
                    else:
                        yield tuple(t if t.is_cuda else t.to(coach.device, non_blocking=True) for t in self._batch(coach, data))

            # one batch ahead: every step is told the next batch (or ticket), which its tail launch prepares (SASRecEngine._train_step_graph_tail)
            for cur, nxt in _lookahead(batches()):
                if isinstance(cur, tuple):
                    eng.train_step_graph(*cur, next_batch=nxt if isinstance(nxt, tuple) else None)
                    n += cur[0].shape[0]
                else:
                    eng.train_step_graph_sampled(cur, next_ticket=nxt if nxt is not None and not isinstance(nxt, tuple) else None)
                    n += len(cur)
            tot = eng.end_loss_accumulation().reshape(())
        eng.check_handover()
        coach.monitor(float(tot / max(n, 1)), n=max(n, 1), reduction="mean", mode="train", pool=["LOSS"])   # (one host read per epoch)
        self.end_epoch(coach)

    def recommend_topk(self, coach, data, seen_ptr, seen_idx, K):
        return self.eng.recommend_topk(data[coach.ISeq].to(coach.device), seen_ptr, seen_idx, K)

    def recommend_pool(self, coach, data):
        return self.eng.recommend_from_pool(data[coach.ISeq].to(coach.device), _pool_tensor(data[coach.IUnseen], coach.device))



class MFAdapter(_Adapter):
    """MF-BPR/main.py:25-131.  The engine's tables are one arena (`User.embeddings.weight | Item.embeddings.weight`)."""
    kind = "MF-BPR"
    NAMES = ("User.embeddings.weight", "Item.embeddings.weight")

    ENGINE = MFEngine

    @classmethod
    def plan(cls, coach, module, spec):
        cls._check_tables(module)
        names = [k for k, _ in module.named_parameters()]
        U, N, D = module.User.count, module.Item.count, module.User.embeddings.weight.shape[1]
        kw = dict(num_users=U, num_items=N, embedding_dim=D, lr=spec.lr, betas=spec.betas, seed=int(coach.cfg.get("seed", 1)))
        kw.update(cls._extra(coach, module, U, N, spec.one_wd(names)))
        return kw

    def __init__(self, coach, module, spec):
        kw = self.plan(coach, module, spec)
        self.module, self.spec = module, spec
        self.eng = self.ENGINE(device=coach.device, **kw)
        self.load_from_module()
        self._tot = torch.zeros((), device=coach.device)

    @classmethod
    def _check_tables(cls, module):
        ok = all(hasattr(module, a) for a in ("User", "Item", "criterion")) and all(
            isinstance(getattr(f, "embeddings", None), nn.Embedding) for f in (module.User, module.Item))
        if not ok or type(module.criterion).__name__ != "BPRLoss" or getattr(module.criterion, "reduction", "mean") != "mean":
            raise Refused("not a two-table BPR model")
        names = sorted(k for k, _ in module.named_parameters())
        if names != sorted(cls.NAMES):
            raise Refused(f"trainable parameters besides the two tables: {names[:4]}")
        Wu, Wi = module.User.embeddings.weight, module.Item.embeddings.weight
        if Wu.shape != (module.User.count, Wi.shape[1]) or Wi.shape[0] != module.Item.count or Wi.shape[1] % 4:
            raise Refused("table shapes")
        if module.User.embeddings.padding_idx is not None or module.Item.embeddings.padding_idx is not None:
            raise Refused("padding rows")

    @staticmethod
    def _extra(coach, module, U, N, wd):
        if hasattr(module, "Adj") or hasattr(module, "num_layers"):
            raise Refused("a graph model, not plain MF")
        return dict(weight_decay=wd)

    def named_views(self):
        return self.eng.params

    def named_grads(self):
        return self.eng.arena.views(self.eng.arena.grad)

    def _batch(self, coach, data):
        dev = coach.device
        return tuple(data[f].to(dev, non_blocking=True).reshape(-1) for f in (coach.User, coach.IPos, coach.INeg))

    def grad_class(self, name):
        return "smooth"                      # (no ReLU anywhere: MF-BPR/main.py:78-93, LightGCN/main.py:77-108)

    def probe_step(self, coach, data):
        u, p, n = self._batch(coach, data)
        if not (u.numel() == p.numel() == n.numel()):
            raise Refused("more than one negative per triplet")
        self.eng.train_step(u, p, n)

    def reset_state(self):
        A = self.eng.arena
        A.m.zero_(); A.v.zero_(); A.step = 0
        if hasattr(self.eng, "_captured"):
            self.eng._captured.clear()

    def wants_fused_sampler(self):
        return False

    def train_epoch(self, coach, epoch):
        self.begin_epoch(coach)
        eng, tot, n = self.eng, self._tot.zero_(), 0
        for data in coach.dataloader:
            u, p, ng = self._batch(coach, data)
            loss = eng.train_step(u, p, ng)          # (eager: the step is 6 launches; its graph replay measured slower, bench_legs config1)
            tot.add_(loss.reshape(()), alpha=u.numel())
            n += u.numel()
        coach.monitor(float(tot / max(n, 1)), n=max(n, 1), reduction="mean", mode="train", pool=["LOSS"])
        self.end_epoch(coach)

    def recommend_topk(self, coach, data, seen_ptr, seen_idx, K):
        return self.eng.recommend_topk(data[coach.User].to(coach.device).reshape(-1), seen_ptr, seen_idx, K)

    def recommend_pool(self, coach, data):
        return self.eng.recommend_from_pool(data[coach.User].to(coach.device).reshape(-1), _pool_tensor(data[coach.IUnseen], coach.device))

    def _adam_wd(self):
        return self.eng.wd


class LightGCNAdapter(MFAdapter):
    """LightGCN/main.py:27-172: CoachForLightGCN builds the optimizer WITHOUT weight decay (:139-145) and differentiates
    rec + cfg.weight_decay * emb (:160) -- the engine's `wd` is that coefficient, its Adam runs with decay 0.  The probe checks both."""
    kind = "LightGCN"

    ENGINE = LightGCNEngine

    @staticmethod
    def _extra(coach, module, U, N, wd):
        Adj = getattr(module, "Adj", None)
        if Adj is None or not hasattr(module, "num_layers") or Adj.layout != torch.sparse_csr or tuple(Adj.shape) != (U + N, U + N):
            raise Refused("no [U + N, U + N] sparse CSR `Adj` buffer / `num_layers`")
        if wd != 0.0:
            raise Refused("LightGCN's optimizer carries a weight decay: not CoachForLightGCN's step")
        return dict(adj_crow=Adj.crow_indices(), adj_col=Adj.col_indices(), adj_val=Adj.values(), num_layers=int(module.num_layers),
                    weight_decay=float(coach.cfg.weight_decay))

    def _adam_wd(self):
        return 0.0


class DeepFMAdapter(_Adapter):
    """DeepFM/main.py:127-276: per-field tables -> the engine's one concatenated table; `dnn.*` by name; the two decay groups of
    `marked_params` (:187-199); clip_grad_norm_(.., 10) (:267); ReduceLROnPlateau stepped on `_best` in front of every epoch (:256)."""
    kind = "DeepFM"

    @classmethod
    def plan(cls, coach, module, spec):
        """-> (engine constructor arguments, fields, name map: module parameter name -> engine key)."""
        if not all(hasattr(module, a) for a in ("input_fields", "dnn", "fm", "criterion", "Label")):
            raise Refused("not DeepFM-shaped")
        if type(module.criterion).__name__ != "BCELoss4Logits" or getattr(module.criterion, "reduction", "mean") != "mean":
            raise Refused(f"criterion {type(module.criterion).__name__}")
        fields = list(module.input_fields)
        for f in fields:
            e, l = getattr(f, "embeddings", None), getattr(f, "embeddings_lr", None)
            if not isinstance(e, nn.Embedding) or not isinstance(l, nn.Embedding) or l.weight.shape[1] != 1 or e.weight.shape[0] != f.count:
                raise Refused(f"field {f.name}: not an (embeddings, embeddings_lr) pair of nn.Embedding (dense fields are not on the engine)")
        blocks = list(module.dnn)
        if not blocks or not isinstance(blocks[-1], nn.Linear) or blocks[-1].out_features != 1:
            raise Refused("dnn does not end in Linear(., 1)")
        hidden, bns, drops = [], set(), set()
        for b in blocks[:-1]:
            if not isinstance(getattr(b, "linear", None), nn.Linear) or not isinstance(getattr(b, "act", None), nn.ReLU):
                raise Refused("dnn block is not Linear -> [BatchNorm1d] -> ReLU -> Dropout")
            hidden.append(b.linear.out_features)
            bns.add(isinstance(getattr(b, "bn", None), nn.BatchNorm1d))
            drops.add(float(b.dropout.p) if isinstance(getattr(b, "dropout", None), nn.Dropout) else 0.0)
            if isinstance(b.bn, nn.BatchNorm1d) and (b.bn.momentum != 0.1 or abs(b.bn.eps - 1e-5) > 0 or not b.bn.affine):
                raise Refused("BatchNorm1d with non-default momentum / eps")
        if len(bns) != 1 or len(drops) != 1:
            raise Refused("dnn blocks differ in batch norm / dropout")
        if not hasattr(module.fm, "lr_layer") or not hasattr(module.fm.lr_layer, "bias"):
            raise Refused("no fm.lr_layer.bias")
        D = fields[0].embeddings.weight.shape[1]
        if blocks[0].linear.in_features != len(fields) * D:
            raise Refused("dnn input width != fields x D")
        vmap, emb, other = cls._name_maps(module, fields)
        kw = dict(counts=[int(f.count) for f in fields], embedding_dim=D, hidden_dims=tuple(hidden), batch_norm=bns == {True},
                  hidden_dropout_rate=drops.pop(), lr=spec.lr, embedding_decay=spec.one_wd(emb), weight_decay=spec.one_wd(other), betas=spec.betas,
                  seed=int(coach.cfg.get("seed", 1)))
        return kw, fields, vmap

    def __init__(self, coach, module, spec):
        kw, self.fields, self._v = self.plan(coach, module, spec)
        self.module, self.spec = module, spec
        self.eng = DeepFMEngine(device=coach.device, **kw)
        self.load_from_module()
        self._bind_running(copy_only=True)
        self.max_norm = 10.0
        self._tot = torch.zeros((), device=coach.device)

    @staticmethod
    def _name_maps(module, fields):
        """module parameter name -> key into the engine (("T", f) / ("TL", f) / engine name); the two decay groups' names."""
        ids = {id(p): n for n, p in module.named_parameters()}
        vmap, emb, other = {}, [], []
        for f, fld in enumerate(fields):
            emb.append(ids[id(fld.embeddings.weight)]); vmap[emb[-1]] = ("T", f)
            emb.append(ids[id(fld.embeddings_lr.weight)]); vmap[emb[-1]] = ("TL", f)
        n = ids[id(module.fm.lr_layer.bias)]
        other.append(n); vmap[n] = "fm.lr_layer.bias"
        dnn_prefix = next(k for k, m in module.named_modules() if m is module.dnn)
        for k, _ in module.dnn.named_parameters():
            other.append(f"{dnn_prefix}.{k}"); vmap[other[-1]] = f"dnn.{k}"
        unknown = [n for n in ids.values() if n not in vmap]
        if unknown:
            raise Refused(f"parameters the DeepFM engine does not have: {unknown[:3]}")
        return vmap, emb, other

    def _views(self, P, T, TL):
        out = {}
        for name, key in self._v.items():
            out[name] = (T if key[0] == "T" else TL)[key[1]] if isinstance(key, tuple) else P[key]
        return out

    def named_views(self):
        e = self.eng
        return self._views(e.P, e.tables(), e.tables_lr())

    def named_grads(self):
        e = self.eng
        off, cnt = e.offsets.tolist(), e.counts
        return self._views(e.G, [e.gT[o:o + c] for o, c in zip(off, cnt)], [e.gTL[o:o + c] for o, c in zip(off, cnt)])

    def _bind_running(self, copy_only=False):
        """BatchNorm running statistics: the module's buffers become the engine's tensors (the script's own eval forward reads them)."""
        for i, b in enumerate(list(self.module.dnn)[:-1]):
            if i in self.eng.running:
                rm, rv = self.eng.running[i]
                if copy_only:
                    rm.copy_(b.bn.running_mean); rv.copy_(b.bn.running_var)
                else:
                    b.bn.running_mean.data, b.bn.running_var.data = rm, rv

    def bind(self):
        super().bind()
        self._bind_running()

    def _batch(self, coach, data):
        dev = coach.device
        x = torch.cat([data[f].to(dev, non_blocking=True).reshape(-1, 1) for f in self.fields], 1).contiguous()
        return x, data[coach.Label].to(dev, non_blocking=True).reshape(-1)

    def grad_class(self, name):
        key = self._v[name]
        if isinstance(key, tuple):
            return "upstream" if key[0] == "T" else "smooth"          # embedding tables feed the MLP; the LR tables see dlogit alone
        nl = self.eng.nl
        if key == "fm.lr_layer.bias" or key.startswith(f"dnn.{nl}."):
            return "smooth"                                           # LR bias; the last layer (its input is ~0 wherever a gate could flip)
        return "gate"                                                 # dnn.i.linear.* / dnn.i.bn.*: directly in front of block i's ReLU

    def probe_step(self, coach, data):
        e = self.eng.train()
        p, e.p_drop = e.p_drop, 0.0
        try:
            e.train_step(*self._batch(coach, data), max_norm=self.max_norm)
        finally:
            e.p_drop = p

    def reset_state(self):
        e = self.eng
        e.m.zero_(); e.v.zero_(); e.step = 0
        self._bind_running(copy_only=True)
        if hasattr(e, "_captured"):
            e._captured.clear()

    def wants_fused_sampler(self):
        return False

    def train_epoch(self, coach, epoch):
        self.begin_epoch(coach)
        e, tot, n, steps = self.eng.train(), self._tot.zero_(), 0, 0
        for data in coach.dataloader:
            x, y = self._batch(coach, data)
            loss = e.train_step(x, y, max_norm=self.max_norm)
            tot.add_(loss.reshape(()), alpha=x.shape[0])
            n += x.shape[0]
            steps += 1
        for b in list(self.module.dnn)[:-1]:
            if isinstance(getattr(b, "bn", None), nn.BatchNorm1d) and b.bn.num_batches_tracked is not None:
                b.bn.num_batches_tracked += steps
        coach.monitor(float(tot / max(n, 1)), n=max(n, 1), reduction="mean", mode="train", pool=["LOSS"])
        self.end_epoch(coach)

    def reset_ranking_buffers(self):
        self.eng.eval()

    def pool_logits(self, coach, data):
        """-> (logits [B], labels [B]) of an evaluation batch (DeepFM/main.py:217-219 returns their sigmoid)."""
        x, y = self._batch(coach, data)
        return self.eng.encode(x)[0], y

    def named_moments(self):
        e = self.eng
        off, cnt = e.offsets.tolist(), e.counts
        M, V = e._views(e.m), e._views(e.v)
        per = lambda X: ([X["T"][o:o + c] for o, c in zip(off, cnt)], [X["TL"][o:o + c] for o, c in zip(off, cnt)])   # noqa: E731
        return self._views(M, *per(M)), self._views(V, *per(V))

    def _get_step(self):
        return int(self.eng.step)

    def _set_step(self, n):
        self.eng.step = int(n)


# ---------------------------------------------------------------------------------------------------------------------------------------
# the probe
# ---------------------------------------------------------------------------------------------------------------------------------------
def _first_batch(coach):
    for data in coach.trainpipe:
        return data
    raise Refused("the training pipe yields no batch to probe with")


class _NoDropout:
    """Dropout off in every module that carries a rate (nn.Dropout.p, nn.MultiheadAttention.dropout)."""

    def __init__(self, module):
        self.saved = []
        for m in module.modules():
            if isinstance(m, (nn.Dropout, nn.Dropout1d, nn.Dropout2d, nn.AlphaDropout)):
                self.saved.append((m, "p", m.p))
            elif isinstance(m, nn.MultiheadAttention):
                self.saved.append((m, "dropout", m.dropout))

    def __enter__(self):
        for m, a, _ in self.saved:
            setattr(m, a, 0.0)

    def __exit__(self, *exc):
        for m, a, v in self.saved:
            setattr(m, a, v)


def _script_step(coach, module, data, dtype=torch.float32):
    """Two iterations of the script's own `train_per_epoch` on `data`; -> (gradients handed to the first optimizer.step(), call trace).
    Module, optimizer, scheduler and monitors come back as they were.  dtype = torch.float64: the module computes in double precision for
    the duration (`module.double()`; the Parameter objects -- what the optimizer holds -- stay the same) -- the script's arithmetic
    without its rounding, which is what an fp32 implementation of it has to be close to."""
    opt, sched = coach.optimizer, getattr(coach, "lr_scheduler", None)
    snap_m = {k: v.detach().clone() for k, v in module.state_dict().items() if v.layout == torch.strided}   # (sparse buffers -- Adj -- are constants)
    snap_o = copy.deepcopy(opt.state_dict())
    snap_s = copy.deepcopy(sched.state_dict()) if sched is not None else None
    meters = {mode: {k: (m.sum, m.n) for k, m in ms.items()} for mode, ms in getattr(coach, "_meters", {}).items()}
    trace, grads = [], {}
    step0, had_step = opt.step, opt.__dict__.get("step")       # (an LRScheduler may already have patched the instance's `step`)

    def step(*a, **k):
        if not grads:
            for n, p in module.named_parameters():
                grads[n] = torch.zeros_like(p) if p.grad is None else p.grad.detach().clone()
        trace.append(("opt",))
        return step0(*a, **k)
    opt.step = step
    if sched is not None:
        sstep0, had_sstep = sched.step, sched.__dict__.get("step")

        def sstep(*a, **k):
            is_best = len(a) == 1 and not k and (a[0] is coach._best or a[0] == coach._best)
            trace.append(("sched", "best" if is_best else "none" if not a and not k else "other"))
            return sstep0(*a, **k)
        sched.step = sstep
    loader, was_training = coach.dataloader, module.training
    try:
        coach.dataloader = [data, data]
        module.train()
        if dtype != torch.float32:
            module.to(dtype)
        with _NoDropout(module), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            type(coach).train_per_epoch(coach, 0)
    finally:
        coach.dataloader = loader
        if had_step is None:
            del opt.step
        else:
            opt.step = had_step
        if sched is not None:
            if had_sstep is None:
                del sched.step
            else:
                sched.step = had_sstep
        if dtype != torch.float32:
            module.to(torch.float32)
        with torch.no_grad():
            for k, v in module.state_dict().items():
                if k in snap_m:
                    v.copy_(snap_m[k])
        module.train(was_training)
        opt.load_state_dict(snap_o)
        opt.state.clear()
        if sched is not None:
            sched.load_state_dict(snap_s)
        for mode, ms in meters.items():
            for k, (s, n) in ms.items():
                coach._meters[mode][k].sum, coach._meters[mode][k].n = s, n
    return grads, trace


def _reference_grads(coach, module, data):
    """-> (reference gradients, the script's fp32 gradients or None, call trace).  The reference is the script's step in DOUBLE precision
    where the module runs in it (everything the four scripts do does); a module that cannot (a hard-coded float32 tensor inside) is
    compared in fp32 at GRAD_TOL_F32 instead."""
    g32, trace = _script_step(coach, module, data)
    try:
        g64, trace64 = _script_step(coach, module, data, torch.float64)
    except Refused:
        raise
    except Exception:  # noqa: BLE001
        return g32, None, trace
    if [t[0] for t in trace64] != [t[0] for t in trace]:
        raise Refused("the script's step loop is not repeatable (different optimizer / scheduler calls on the same batch)")
    return g64, g32, trace


def _replayable(trace):
    """-> sched_mode for a call trace of a two-batch epoch, or raises."""
    opts = [i for i, t in enumerate(trace) if t[0] == "opt"]
    sch = [(i, t[1]) for i, t in enumerate(trace) if t[0] == "sched"]
    if len(opts) != 2:
        raise Refused(f"{len(opts)} optimizer steps for two batches (gradient accumulation?)")
    if not sch:
        return None
    if len(sch) == 1 and sch[0][0] < opts[0] and sch[0][1] == "best":
        return "front_best"
    if len(sch) == 1 and sch[0][0] > opts[-1] and sch[0][1] == "none":
        return "back"
    raise Refused("a learning-rate schedule the adapters do not replay (stepped per batch, or with other arguments)")


def _probe(coach, ad):
    module = ad.module
    data = _first_batch(coach)
    data = {k: (v.to(coach.device) if torch.is_tensor(v) else v) for k, v in data.items()}
    before = {k: v.detach().clone() for k, v in ad.named_views().items()}
    g_ref, g_f32, trace = _reference_grads(coach, module, data)
    ad.sched_mode = _replayable(trace)
    ad.probe_step(coach, data)
    torch.cuda.synchronize()
    lr, (b1, b2) = ad.spec.lr, ad.spec.betas
    views, grads = ad.named_views(), ad.named_grads()
    worst, first = ("", 0.0), {}
    # the model's gradient as a whole: a tensor whose gradient is a millionth of it is rounding residue in BOTH arithmetics (a Linear bias in
    # front of a BatchNorm: mathematically zero, 1e-19 in double precision, 1e-10 .. 1e-9 in either fp32 form -- and the script's own residue,
    # measured once, is no bound on the engine's: 6.5e-10 against 4 x 1.5e-10 refused a DeepFM in about one fresh process of six)
    g_all_max = max(float(g_ref[k].abs().max()) for k in views)
    g_all_n2 = math.sqrt(sum(float(g_ref[k].double().norm()) ** 2 for k in views))
    for k, v in views.items():
        ge, gr = grads[k].detach().reshape(-1).double(), g_ref[k].reshape(-1).double()
        scale = float(gr.abs().max())
        err = float((ge - gr).abs().max())
        # allowed: GRAD_TOL in the L2 sense and 10 x GRAD_TOL for the worst entry (a relu pre-activation within rounding of zero may fall on the
        # other side in one of two fp32 arithmetics; that unit's gradient entries then move by its whole contribution -- a handful among
        # ~10^5 - 10^6 pre-activations of a batch) -- or, where the script's OWN fp32 arithmetic is noisier than that (sums that cancel: a
        # Linear bias behind a BatchNorm has a zero gradient; ROCm aten's fp32 GEMMs), a few times the script's own distance from its
        # double-precision self.  Different arithmetic (another mask, another normalisation, a missing term) is off by O(1) in both norms.
        tol_rel = (GRAD_TOL if g_f32 is not None else GRAD_TOL_F32) if ad.grad_class(k) == "gate" else (
            GRAD_TOL_SMOOTH if ad.grad_class(k) == "smooth" else GRAD_TOL_UPSTREAM)
        d32 = None if g_f32 is None else (g_f32[k].reshape(-1).double() - gr)
        noise = 0.0 if d32 is None else float(d32.abs().max())
        noise2 = 0.0 if d32 is None else float(d32.norm())
        err2, n2 = float((ge - gr).norm()), float(gr.norm())
        tol = max(10.0 * tol_rel * scale, 4.0 * noise, 1e-6 * g_all_max) + 1e-12
        tol2 = max(tol_rel * n2, 4.0 * noise2, 1e-6 * g_all_n2) + 1e-12
        if os.environ.get("RECENGINE_PROBE_REPORT"):      # (diagnostic: every tensor's distance, scripts/probe_margin.py)
            print(f"PROBE {type(module).__name__} {k}: L2 {err2 / max(n2, 1e-300):.3e} of the tensor's, worst entry {err / max(scale, 1e-300):.3e} of its largest, "
                  f"noise x4 L2 {4 * noise2 / max(n2, 1e-300):.1e}", flush=True)
        if not math.isfinite(err) or err > tol or err2 > tol2:
            raise Refused(f"gradient of {k} differs from the script's own step: |diff| max {err:.3e} / L2 {err2:.3e} against |grad| max {scale:.3e} / L2 "
                          f"{n2:.3e} (the script's own fp32 rounding there: max {noise:.1e})")
        if scale > 0 and err / scale > worst[1]:
            worst = (k, err / scale)
        # torch.optim.Adam's first step on the ENGINE's gradient with the script's numbers (coupled decay, eps 1e-8)
        p0 = before[k].reshape(-1).double()
        g = ge + ad.spec.wd[k] * p0
        want = -lr * g / (g.abs() + 1e-8)
        got = v.detach().reshape(-1).double() - p0
        uerr = float((got - want).abs().max())
        if not math.isfinite(uerr) or uerr > UPDATE_TOL * lr + 2.4e-7 * float(p0.abs().max()) + 1e-12:   # (+ two ulps of the largest parameter)
            raise Refused(f"update of {k} is not Adam(lr={lr}, betas=({b1}, {b2}), weight_decay={ad.spec.wd[k]}) on its gradient: off by {uerr:.3e}")
        first[k] = (ge, v.detach().reshape(-1).double().clone())
    # A SECOND step on the same batch: the first Adam update is -lr g / (|g| + eps) whatever the betas are; the second one,
    # m2 / (1 - b1^2) over sqrt(v2 / (1 - b2^2)) + eps with m2 = b1 (1 - b1) g1 + (1 - b1) g2 and v2 likewise, shows whether the engine runs
    # the SCRIPT's (beta1, beta2) and its own step count.
    ad.probe_step(coach, data)
    torch.cuda.synchronize()
    views, grads = ad.named_views(), ad.named_grads()
    for k, v in views.items():
        g1, p1 = first[k]
        g1 = g1 + ad.spec.wd[k] * before[k].reshape(-1).double()
        g2 = grads[k].detach().reshape(-1).double() + ad.spec.wd[k] * p1
        m2 = (b1 * (1 - b1) * g1 + (1 - b1) * g2) / (1 - b1 ** 2)
        v2 = (b2 * (1 - b2) * g1 * g1 + (1 - b2) * g2 * g2) / (1 - b2 ** 2)
        want = -lr * m2 / (v2.sqrt() + 1e-8)
        got = v.detach().reshape(-1).double() - p1
        uerr = float((got - want).abs().max())
        if not math.isfinite(uerr) or uerr > UPDATE_TOL * lr + 2.4e-7 * float(p1.abs().max()) + 1e-12:
            raise Refused(f"second update of {k} is not Adam's with betas=({b1}, {b2}) on its two gradients: off by {uerr:.3e} (lr {lr})")
    # back to the parameters the script handed over
    with torch.no_grad():
        for k, v in views.items():
            v.copy_(before[k])
    ad.reset_state()
    return worst


def candidate(m):
    """The adapter class whose model `m` LOOKS like (by the names of its parts), or None."""
    if hasattr(m, "attnLayers") and hasattr(m, "Position"):
        return SASRecAdapter
    if hasattr(m, "input_fields") and hasattr(m, "dnn") and hasattr(m, "fm"):
        return DeepFMAdapter
    if hasattr(getattr(m, "User", None), "embeddings") and hasattr(getattr(m, "Item", None), "embeddings"):
        return LightGCNAdapter if hasattr(m, "Adj") else MFAdapter
    return None


def _preconditions(coach):
    from freerec import ddp, launcher
    if ddp.is_distributed():
        raise Refused("running under torchrun: the engines' captured steps carry no gradient all-reduce")
    if type(coach).train_per_epoch is launcher.Coach.train_per_epoch:
        raise Refused("the Coach defines no train_per_epoch to compare with")


def survey(coach, data=None):
    """Everything of the adoption decision that needs no GPU (tests/test_bridge_host.py runs it on the reference's scripts imported in place):
    -> (adapter class, OptSpec, engine constructor plan, the script's step gradients on `data` (default: the first training batch), the
    scheduler replay mode).  Raises Refused."""
    m = coach.get_res_sys_arch()
    cls = candidate(m)
    if cls is None:
        raise Refused("no engine for this model")
    _preconditions(coach)
    spec = OptSpec(coach, m)
    plan = cls.plan(coach, m, spec)
    data = _first_batch(coach) if data is None else data
    grads, _, trace = _reference_grads(coach, m, {k: (v.to(coach.device) if torch.is_tensor(v) else v) for k, v in data.items()})
    return cls, spec, plan, {k: g.to(torch.float32) for k, g in grads.items()}, _replayable(trace)


def attach(coach):
    """-> an adapter when `coach.model` is a model the engine runs fused AND the probe shows the engine's step is the script's; else None
    (with a logged reason when the model looked like a candidate)."""
    m = coach.get_res_sys_arch()
    cls = candidate(m)
    if cls is None:
        return None
    name = type(m).__name__
    try:
        _preconditions(coach)
        spec = OptSpec(coach, m)
        ad = cls(coach, m, spec)
        worst = _probe(coach, ad)
        ad.bind()
        _log(f"[recengine] >>> {name}: training and evaluation run on the {ad.kind} engine (probe step: gradients equal the script's own to "
             f"{worst[1]:.1e} of max at `{worst[0]}`; Adam lr={spec.lr} betas={spec.betas}; scheduler replay: {ad.sched_mode})")
        return ad
    except Refused as e:
        msg = f"[recengine] >>> {name} stays on its own torch code: {e}"
    except Exception as e:  # noqa: BLE001  (anything unexpected about the module: leave it to its own torch code, loudly)
        msg = f"[recengine] >>> {name} stays on its own torch code: {type(e).__name__}: {e}"
    warnings.warn(msg)
    _log(msg)
    return None
