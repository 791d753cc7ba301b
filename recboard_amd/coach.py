"""The Coach for ENGINE OBJECTS driven without a script (bench.py, tests, scripts): a constructor shim over the one Coach of this repo,
`freerec.launcher.Coach` (the reference's `Coach(dataset=, trainpipe=, validpipe=, testpipe=, model=, cfg=).fit()`, SASRec/main.py:278-286;
loop shape evidenced by ETEGRec/train_etegrec.py:625-650; evaluate contract by UniSRec/main.py:400-447).  The epoch loop, the fused
evaluation, best tracking, `checkpoint.tar` / `best.pt` and the leaderboard record (`results.json`) are the base class's; what is here is
(a) the keyword constructor (no dataset / cfg objects), (b) `_Direct`: the "engine adapter" the base class drives, stepping the engine object
itself instead of an adopted script, (c) the host-to-device batch staging of a pipe that hands out CPU tensors.

Per epoch: `train_per_epoch` (SASRec/main.py:242-258 / MF-BPR/main.py:115-131: one engine `train_step` per batch, LOSS
monitored as the mean over batches weighted by batch size); every `eval_freq` epochs `evaluate("valid")` with the
fused score+mask+top-K kernel and the metrics kernel; best epoch tracked on `which4best`.  The per-step `loss.item()`
host sync of the reference is replaced by one device-side accumulation read at the end of the epoch.
"""
import os
import weakref

import torch

from freerec import launcher as _launcher

from .evaluate import PredictionEvaluator, RankingEvaluator, ReduceLROnPlateau, ragged_to_csr  # noqa: F401


def _lookahead(it):
    """(item, next item or None) pairs."""
    it = iter(it)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


class _Cfg(dict):
    """The few cfg entries the base Coach reads, as attributes and through `get` / `to_dict`."""
    __getattr__ = dict.get

    def to_dict(self):
        return dict(self)


class _Direct:
    """What `freerec.launcher.Coach` asks of an engine adapter, for an engine OBJECT (no script, no probe: there is nothing to adopt)."""

    def __init__(self, coach):
        # (a weak reference: coach -> adapter -> coach would be a cycle, and a Coach -- with its engine's captured hipGraphs -- would then be
        #  freed by the cyclic collector at an arbitrary later moment, e.g. in the middle of ANOTHER engine's stream capture, which aborts)
        self.c = weakref.proxy(coach)
        self.is_pred = coach.kind == "pred"

    def __getattr__(self, name):
        # the base class tells a prediction model by the presence of `pool_logits`
        if name == "pool_logits" and self.__dict__.get("is_pred"):
            return self._pool_logits
        raise AttributeError(name)

    def wants_fused_sampler(self):
        return False

    def train_epoch(self, coach, epoch):
        coach.train_per_epoch(epoch)

    def reset_ranking_buffers(self):
        m = self.c.model
        if hasattr(m, "eval"):
            m.eval()
        if hasattr(m, "reset_ranking_buffers"):
            m.reset_ranking_buffers()

    def recommend_topk(self, coach, data, seen_ptr, seen_idx, K):
        key = {"seq": "ISeq", "gen": "User", "module": (coach.fit_keys or ("User",))[0]}[coach.kind]
        return coach.model.recommend_topk(data[key].to(coach.device), seen_ptr, seen_idx, K)

    def recommend_pool(self, coach, data):
        key = {"seq": "ISeq", "gen": "User", "module": (coach.fit_keys or ("User",))[0]}[coach.kind]
        pool = data["IUnseen"]
        pool = pool if torch.is_tensor(pool) else torch.as_tensor(pool, dtype=torch.int64)
        x = data[key].to(coach.device)
        return coach.model.recommend_from_pool(x if coach.kind == "seq" else x.reshape(-1), pool.to(coach.device).reshape(pool.shape[0], -1))

    def _pool_logits(self, coach, data):
        data = coach.dict_to_device(data)
        return coach.model.encode(data["X"])[0], data["Label"]

    def optimizer_state(self, coach):
        m = coach.model
        if hasattr(m, "optimizer_state"):            # engines whose Adam state is more than the arena (large / sharded tables)
            return m.optimizer_state()
        if hasattr(m, "arena"):                      # a torch.optim.Adam-shaped state_dict over the reference's parameter names
            return m.arena.adam_state_dict(m.lr, m.betas, m.wd)
        if hasattr(m, "adam_state_dict"):
            return m.adam_state_dict()
        if coach.optimizer is not None:              # kind = "module": the torch optimizer's own state_dict
            return coach.optimizer.state_dict()
        return {}

    def load_optimizer_state(self, coach, opt):
        m = coach.model
        if hasattr(m, "load_optimizer_state"):
            m.load_optimizer_state(opt)
        elif hasattr(m, "arena"):
            m.arena.load_adam_state_dict(opt)
        elif hasattr(m, "load_adam_state_dict"):
            m.load_adam_state_dict(opt)
        elif coach.optimizer is not None:
            coach.optimizer.load_state_dict(opt)


class Coach(_launcher.Coach):
    User, Item, ISeq, IPos, INeg, IUnseen, ISeen, Label, Size = "User", "Item", "ISeq", "IPos", "INeg", "IUnseen", "ISeen", "Label", "Size"

    def __init__(self, model, trainpipe, validpipe=None, testpipe=None, monitors=("LOSS", "HitRate@10", "NDCG@10"),
                 which4best="NDCG@10", eval_freq=5, kind="seq", checkpoint_path=None, lr_scheduler=None, optimizer=None,
                 fit_keys=None, loss_fn=None, graph=False, dataset=None, pred_metrics="global", seed=0):
        """kind: "seq" (SASRec: data ISeq / IPos / INeg), "gen" (MF-BPR / LightGCN: User / IPos / INeg), "module" (a torch.nn.Module on
        the custom-op surface, recboard_amd.siblings: `model.fit(*[data[k] for k in fit_keys])` -> dict of losses, `loss_fn(losses)` -> the
        scalar to differentiate (default: their sum; e.g. CoachForLightGCN's rec + weight_decay * emb, LightGCN/main.py:156-172),
        `optimizer` a torch optimizer (capturable for graph=True: the step replayed as one hipGraph, nn.GraphedStep); evaluation through
        `model.recommend_topk(data[fit_keys[0]], seen, K)`) or "pred" (DeepFM: a field matrix `X` [B, F] and `Label`; monitors LOGLOSS / AUC,
        DeepFM/configs/Frappe_x1_BARS.yaml:101-102; pred_metrics = "global": ONE AUC / LOGLOSS over the split's rows -- "batch" is the
        reference Coach's batch-weighted mean of per-batch values).  lr_scheduler: e.g. `ReduceLROnPlateau(model, mode="max",
        patience=eval_freq, ...)`, stepped on the best monitored value at the top of every epoch as CoachForDeepFM does (DeepFM/main.py:251-257)."""
        self.kind, self.checkpoint_path = kind, checkpoint_path
        self._given = (optimizer, lr_scheduler)
        self.fit_keys, self.graph = tuple(fit_keys) if fit_keys else None, graph
        self.loss_fn = loss_fn if loss_fn is not None else (lambda losses: sum(losses.values()))
        self._graphed = {}
        if kind == "module" and (optimizer is None or not self.fit_keys):
            raise ValueError("Coach(kind='module') needs `optimizer` and `fit_keys`")
        mons = list(monitors)
        # which4best: a monitor that exists for this run (a training-only Coach monitors LOSS alone)
        has = {m.upper() for m in mons}
        cfg = _Cfg(monitors=mons, which4best=which4best if which4best.upper() in has else mons[0], eval_freq=eval_freq, epochs=0, ranking="full",
                   retain_seen=False, eval_valid=True, eval_test=False, checkpoint_path=checkpoint_path or "", engine="direct", seed=seed,
                   device=str(model.device if hasattr(model, "device") else next(model.parameters()).device), dataset=dataset,
                   description=type(model).__name__, pred_metrics=pred_metrics, checkpoint_freq=1 if checkpoint_path else 0)
        super().__init__(dataset=dataset, trainpipe=trainpipe, validpipe=validpipe, testpipe=testpipe, model=model, cfg=cfg)

    # ---- the base class's set-up hooks
    def set_device(self):
        self.device = torch.device(self.cfg.device)

    def set_model(self, model):
        self.model = model

    def set_optimizer(self):
        self.optimizer = self._given[0]

    def set_lr_scheduler(self):
        self.lr_scheduler = self._given[1]

    def _attach_engine(self):
        return _Direct(self)

    def _saves_files(self):
        return bool(self.checkpoint_path)

    def _set_training(self, flag):
        m = self.model
        if flag and hasattr(m, "train"):
            m.train()
        elif not flag and hasattr(m, "eval"):
            m.eval()

    def _evaluate(self, epoch, step, mode):
        if not self._fused_eval(mode):
            raise NotImplementedError(f"Coach(kind={self.kind!r}): a monitor of {sorted(self._meters[mode])} is not one the evaluation kernels compute")

    def dict_to_device(self, data, keys=None):
        """Coach.dict_to_device: tensors to the model's device (asynchronously when the pipe hands out pinned memory); `keys` limits
        the copies to what the step reads."""
        return {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) and (keys is None or k in keys) else v)
                for k, v in data.items()}

    # ---- the previous interface of this class (tests, bench, scripts)
    @property
    def best(self):
        return None if self._best in (float("inf"), -float("inf")) else (self._best_epoch, self._best)

    def evaluate(self, mode="valid"):
        """-> {monitor: value} of one split under the current parameters (no best tracking, no history)."""
        was = getattr(self.model, "training", False)
        with torch.no_grad():
            self.dataloader = self.validpipe if mode == "valid" else self.testpipe
            self._set_training(False)
            self._evaluate(0, -1, mode)
        if hasattr(self.model, "train"):
            self.model.train(was)
        out = {}
        for name, meter in self._meters[mode].items():
            if meter.n:
                out[name] = meter.avg
            meter.reset()
        return out

    def fit(self, epochs=None):
        if epochs is not None:
            self.cfg["epochs"] = int(epochs)
        sch = self.lr_scheduler
        if sch is not None and getattr(sch, "mode", None) not in (None, "max" if self._best_caster() is max else "min"):
            import warnings
            warnings.warn(f"lr_scheduler.mode={sch.mode!r} but which4best={self.cfg.which4best!r} is {'max' if self._best_caster() is max else 'min'}imised: "
                          "the scheduler will read every improvement as a bad epoch")
        out = super().fit()
        out["best_monitors"], out["best"] = out["best"], self.best
        return out

    def _device_batches(self, pipe, keys, ahead=1):
        """The pipe's batches on the device, `ahead` BATCHES AHEAD: the host-to-device copies of batch i+1 run on a copy stream while step i
        computes (three pinned 200 KB copies are ~70 us of DMA latency per step -- half a training step -- when they share the
        compute stream).  One cross-stream event per step.  ahead=2 for a consumer that itself looks one batch ahead (a pipelined step reads
        batch i+1 during step i: its copies must have been started a step earlier, or the compute stream waits for them)."""
        if self.device.type != "cuda":
            for data in pipe:
                yield self.dict_to_device(data, keys)
            return
        main = torch.cuda.current_stream()
        if not hasattr(self, "_copy_streams"):
            self._copy_streams = [torch.cuda.Stream() for _ in range(3)]

        def stage(data):
            # one copy stream per tensor (round robin): a pinned 200 KB copy is ~40 us of latency, three in a row on ONE stream were 120 us per
            # batch -- more than a training step, and what an epoch from host batches ran at
            d, used = {}, []
            j = 0
            # same-shape int64 tensors (a sequence batch: ISeq / IPos / INeg) travel as ONE copy: packed into a pinned staging buffer of a
            # ring (a host memcpy), one transfer, views on the device
            pack = [k for k, v in data.items() if isinstance(v, torch.Tensor) and (keys is None or k in keys) and v.device.type == "cpu"
                    and v.dtype == torch.int64 and v.dim() == 2]
            if getattr(self, "pack_copies", True) and len(pack) >= 2 and len({tuple(data[k].shape) for k in pack}) == 1:
                shp = tuple(data[pack[0]].shape)
                ring = self.__dict__.setdefault("_pack_ring", {})
                slot = ring.get((len(pack),) + shp)
                if slot is None:
                    slot = ring[(len(pack),) + shp] = {"bufs": [torch.empty((len(pack),) + shp, dtype=torch.int64).pin_memory()
                                                                for _ in range(ahead + 3)], "done": [None] * (ahead + 3), "i": 0}
                bi = slot["i"] % len(slot["bufs"])
                slot["i"] += 1
                buf = slot["bufs"][bi]
                if slot["done"][bi] is not None:
                    slot["done"][bi].synchronize()             # (the transfer that last read this staging buffer: long done unless the host runs far ahead)
                for n, k in enumerate(pack):
                    buf[n].copy_(data[k])
                st = self._copy_streams[0]
                with torch.cuda.stream(st):
                    dev = buf.to(self.device, non_blocking=True)
                    slot["done"][bi] = torch.cuda.Event()
                    slot["done"][bi].record(st)
                used.append(st)
                for n, k in enumerate(pack):
                    d[k] = dev[n]
                data = {k: v for k, v in data.items() if k not in pack}
            for k, v in data.items():
                if isinstance(v, torch.Tensor) and (keys is None or k in keys) and v.device != self.device:
                    st = self._copy_streams[j % len(self._copy_streams)]
                    j += 1
                    with torch.cuda.stream(st):
                        d[k] = v.to(self.device, non_blocking=True)
                    if st not in used:
                        used.append(st)
                else:
                    d[k] = v
            evs = []
            for st in used:
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            return d, evs

        import collections
        it = iter(pipe)
        q = collections.deque()

        def fill():
            while len(q) < ahead:
                try:
                    q.append(stage(next(it)))
                except StopIteration:
                    return False
            return True

        # (a cross-stream wait in front of a step costs ~10 us on this stack: with several batches staged ahead the compute stream waits
        #  for the NEWEST staged batch's event once per group of `group` batches -- the copy stream is in order, so that covers them all)
        group = 4 if ahead >= 2 else 1
        covered = 0
        more = fill()
        while q:
            cur, ev = q.popleft()
            if covered > 0:
                covered -= 1
            else:
                n_cov = min(group - 1, len(q))
                for e in (q[n_cov - 1][1] if n_cov > 0 else ev):
                    main.wait_event(e)
                covered = n_cov
            if more:
                more = fill()
            for v in cur.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(main)
            yield cur

    def train_per_epoch(self, epoch):
        if self.lr_scheduler is not None:            # DeepFM/main.py:256: self.lr_scheduler.step(self._best)
            self.lr_scheduler.step(self._best)
        tot = torch.zeros((), device=self.device)
        n = 0
        need = {"seq": ("ISeq", "IPos", "INeg"), "pred": ("X", "Label"), "module": self.fit_keys}.get(self.kind)
        pipelined = self.kind == "seq" and self._graphable() and (getattr(self.model, "pipelined_prep", False) or
                                                                 (getattr(self.model, "_tail_prep_ok", None) is not None and self.model._tail_prep_ok()))
        batches = self._device_batches(self.trainpipe, need, ahead=8 if pipelined else 1)
        # the fused SASRec step sums the epoch's losses itself (each step's loss is folded in by the next step's preparation launch)
        own_sum = self.kind == "seq" and self._graphable() and hasattr(self.model, "begin_loss_accumulation")
        if own_sum:
            self.model.begin_loss_accumulation()
        if pipelined:
            batches = _lookahead(batches)
        for data in batches:
            if pipelined:
                data, nxt = data
            if pipelined and "Sample" not in data:
                # the NEXT batch is already on the device (one batch ahead): it is prepared during this step
                loss = self.model.train_step_graph(data["ISeq"], data["IPos"], data["INeg"],
                                                   next_batch=None if nxt is None or "Sample" in nxt else (nxt["ISeq"], nxt["IPos"], nxt["INeg"]))
                bsz = len(data["User"])
                if not own_sum:
                    tot.add_(loss, alpha=bsz)
                n += bsz
                continue
            if self.kind == "seq" and "Sample" in data:       # a fused device sampler's ticket: the step's preparation launch samples the batch
                if pipelined and nxt is not None and "Sample" in nxt:   # ... or, one ticket ahead, the PREVIOUS step's tail launch did
                    loss = self.model.train_step_graph_sampled(data["Sample"], next_ticket=nxt["Sample"])
                else:
                    loss = self.model.train_step_graph_sampled(data["Sample"])
                bsz = len(data["Sample"])
                if not own_sum:
                    tot.add_(loss, alpha=bsz)
                n += bsz
                continue
            if self.kind == "module":
                loss = self._module_step(tuple(data[k] for k in self.fit_keys))
                bsz = len(data[self.fit_keys[0]])
                tot.add_(loss.detach(), alpha=bsz)
                n += bsz
                continue
            if self.kind == "pred":                  # DeepFM/main.py:258-268: forward, backward, clip_grad_norm_(.., 10), step
                loss = self.model.train_step(data["X"], data["Label"])
                bsz = data["X"].shape[0]
                tot += loss * bsz
                n += bsz
                continue
            if self.kind == "seq" and self._graphable():
                # one batch-preparation launch + one hipGraph replay per step (SASRecEngine.train_step_graph); a short last batch
                # gets its own captured graph.  (The preparation launch one batch ahead on the copy stream was measured: 0.305 vs
                # 0.175 ms per step -- its buffer-reuse wait holds the next copies back.)
                loss = self.model.train_step_graph(data["ISeq"], data["IPos"], data["INeg"])
            elif self.kind == "seq":
                loss = self.model.train_step(data["ISeq"], data["IPos"], data["INeg"])
            else:
                loss = self.model.train_step(data["User"], data["IPos"], data["INeg"])
            bsz = len(data["User"])
            if not (own_sum and self.kind == "seq" and self._graphable()):
                tot.add_(loss, alpha=bsz)            # (one launch)
            n += bsz
        if own_sum:
            tot = self.model.end_loss_accumulation().reshape(())
        if hasattr(self.model, "check_handover"):
            self.model.check_handover()     # (split long sequences: the halves' hand-over flags; the loss read below syncs anyway)
        if hasattr(self.model, "settle_overflow"):
            self.model.settle_overflow()    # (row-sharded tables: steps whose exchange overflowed were no-ops; they are re-run here at the latest)
        table = getattr(self.model, "table", None)
        if table is not None and hasattr(table, "check_capacity"):
            table.check_capacity()          # (row-sharded tables: a lookup dropped by a full exchange bucket came back as a zero row)
        loss = float(tot / max(n, 1))                # (one host read per epoch)
        self.monitor(loss, n=max(n, 1), reduction="mean", mode="train", pool=["LOSS"])
        return {"LOSS": loss}

    def _module_step(self, inputs):
        """One optimizer step of a torch.nn.Module model: eager, or (graph=True) the whole step replayed as one hipGraph per input shape."""
        m = self.model
        if not self.graph:
            self.optimizer.zero_grad(set_to_none=True)
            loss = self.loss_fn(m.fit(*inputs))
            loss.backward()
            self.optimizer.step()
            return loss
        from .nn import GraphedStep
        key = tuple((tuple(t.shape), t.dtype) for t in inputs)
        if hasattr(m, "static_shapes") or type(m).__name__ in ("GRU4Rec", "NARM", "BERT4Rec"):
            m.static_shapes = True          # (their reference forms drop all-pad columns / index by a mask: data-dependent shapes, a host sync)
        if key not in self._graphed:
            self._graphed[key] = GraphedStep(m, lambda *a: self.loss_fn(m.fit(*a)), self.optimizer, inputs)
        return self._graphed[key](*inputs)


    def _graphable(self):
        m = self.model
        return hasattr(m, "train_step_graph") and getattr(m, "encoder", None) == "fused" and getattr(m, "loss_kind", "CE") != "CE"
