"""A compact Coach for the engine models: host-side mirror of `freerec.launcher.Coach` as the reference scripts use it
(`Coach(dataset=, trainpipe=, validpipe=, testpipe=, model=, cfg=).fit()`, SASRec/main.py:278-286; loop shape evidenced by
ETEGRec/train_etegrec.py:625-650; evaluate contract by UniSRec/main.py:400-447).

Per epoch: `train_per_epoch` (SASRec/main.py:242-258 / MF-BPR/main.py:115-131: one engine `train_step` per batch, LOSS
monitored as the mean over batches weighted by batch size); every `eval_freq` epochs `evaluate("valid")` with the
fused score+mask+top-K kernel and the metrics kernel; best epoch tracked on `which4best`.  The per-step `loss.item()`
host sync of the reference is replaced by one device-side accumulation read at the end of the epoch.
"""
import datetime
import json
import os

import torch

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


class Coach:
    def __init__(self, model, trainpipe, validpipe=None, testpipe=None, monitors=("LOSS", "HitRate@10", "NDCG@10"),
                 which4best="NDCG@10", eval_freq=5, kind="seq", checkpoint_path=None, lr_scheduler=None, optimizer=None,
                 fit_keys=None, loss_fn=None, graph=False):
        """kind: "seq" (SASRec: data ISeq / IPos / INeg), "gen" (MF-BPR / LightGCN: User / IPos / INeg), "module" (a torch.nn.Module on
        the custom-op surface, recboard_amd.siblings: `model.fit(*[data[k] for k in fit_keys])` -> dict of losses, `loss_fn(losses)` -> the
        scalar to differentiate (default: their sum; e.g. CoachForLightGCN's rec + weight_decay * emb, LightGCN/main.py:156-172),
        `optimizer` a torch optimizer (capturable for graph=True: the step replayed as one hipGraph, nn.GraphedStep); evaluation through
        `model.recommend_topk(data[fit_keys[0]], seen, K)`) or "pred" (DeepFM: a field matrix
        `X` [B, F] and `Label`; monitors LOGLOSS / AUC, DeepFM/configs/Frappe_x1_BARS.yaml:101-102).  lr_scheduler: e.g.
        `ReduceLROnPlateau(model, mode="max", patience=eval_freq, ...)`, stepped on the best monitored value at the top of every
        epoch as CoachForDeepFM does (DeepFM/main.py:251-257)."""
        self.model, self.trainpipe, self.validpipe, self.testpipe = model, trainpipe, validpipe, testpipe
        self.monitors, self.which4best, self.eval_freq, self.kind = list(monitors), which4best, eval_freq, kind
        self.history, self.best = [], None
        self.checkpoint_path = checkpoint_path
        self.device = model.device if hasattr(model, "device") else next(model.parameters()).device
        self.lr_scheduler = lr_scheduler
        self.optimizer, self.fit_keys, self.graph = optimizer, tuple(fit_keys) if fit_keys else None, graph
        self.loss_fn = loss_fn if loss_fn is not None else (lambda losses: sum(losses.values()))
        self._graphed = {}
        if kind == "module" and (optimizer is None or not self.fit_keys):
            raise ValueError("Coach(kind='module') needs `optimizer` and `fit_keys`")

    def dict_to_device(self, data, keys=None):
        """Coach.dict_to_device: tensors to the model's device (asynchronously when the pipe hands out pinned memory); `keys` limits
        the copies to what the step reads."""
        return {k: (v.to(self.device, non_blocking=True) if isinstance(v, torch.Tensor) and (keys is None or k in keys) else v)
                for k, v in data.items()}

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
            self.lr_scheduler.step(self.best[1] if self.best is not None else (-float("inf") if self.lr_scheduler.mode == "max" else float("inf")))
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
        return {"LOSS": float(tot / max(n, 1))}

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

    # ---- checkpoint / results in the reference's formats: `checkpoint.tar` (model state_dict under the
    #      reference's parameter names + optimizer state + epoch), `best.pt` (state_dict of the best epoch), and the
    #      `benchmark/<dataset>/<model>.json` record schema (benchmark/Amazon2014Beauty_550_LOU/SASRec.json:1-304).
    def _optimizer_state(self):
        m = self.model
        if hasattr(m, "optimizer_state"):            # engines whose Adam state is more than the arena (large / sharded tables)
            return m.optimizer_state()
        if hasattr(m, "arena"):                      # a torch.optim.Adam-shaped state_dict over the reference's parameter names
            return m.arena.adam_state_dict(m.lr, m.betas, m.wd)
        if hasattr(m, "adam_state_dict"):
            return m.adam_state_dict()
        if self.optimizer is not None:               # kind = "module": the torch optimizer's own state_dict
            return self.optimizer.state_dict()
        return {}

    def save_checkpoint(self, path, epoch):
        """`checkpoint.tar` with the reference's keys (freerec Coach.save_checkpoint, shape evidenced by ETEGRec/train_etegrec.py:549-574
        and the cfg dump's CHECKPOINT_MODULES): {epoch, model, optimizer, lr_scheduler, monitors}."""
        os.makedirs(path, exist_ok=True)
        torch.save({"epoch": epoch, "model": self.model.state_dict(), "optimizer": self._optimizer_state(),
                    "lr_scheduler": self.lr_scheduler.state_dict() if self.lr_scheduler is not None else None,
                    "monitors": {"best": self.best, "history": self.history}},
                   os.path.join(path, "checkpoint.tar"))

    def load_checkpoint(self, path):
        ck = torch.load(os.path.join(path, "checkpoint.tar"), map_location=self.device, weights_only=False)
        m = self.model
        m.load_state_dict(ck["model"])
        opt = ck.get("optimizer") or {}
        if opt:
            if hasattr(m, "load_optimizer_state"):
                m.load_optimizer_state(opt)
            elif hasattr(m, "arena"):
                m.arena.load_adam_state_dict(opt)
            elif hasattr(m, "load_adam_state_dict"):
                m.load_adam_state_dict(opt)
            elif self.optimizer is not None:
                self.optimizer.load_state_dict(opt)
        if self.lr_scheduler is not None and ck.get("lr_scheduler"):
            self.lr_scheduler.load_state_dict(ck["lr_scheduler"])
        mon = ck.get("monitors") or {}           # (checkpoints written before `monitors` existed keep best / history at the top level)
        self.best, self.history = mon.get("best", ck.get("best")), mon.get("history", ck.get("history", []))
        return ck["epoch"]

    def save_best(self, path):
        os.makedirs(path, exist_ok=True)
        torch.save(self.model.state_dict(), os.path.join(path, "best.pt"))

    def results_record(self, dataset, model_name, out, seed=0, config=None, run_id=None):
        """-> the list-of-one record the leaderboard's build-data script reads (recboard/scripts/build-data.mjs:95-146)."""
        last_train = self.history[-1]["train"] if self.history else {}
        best_valid = {}
        if self.best is not None:
            best_valid = next((h["valid"] for h in self.history if h["epoch"] == self.best[0] and "valid" in h), {})
        now = datetime.datetime.now()
        return [{
            "description": "", "dataset": dataset, "tags": [model_name, "recengine", "MI355X"],
            "runs": [{"id": run_id or now.strftime("%m%d%H%M%S"), "params": {"config": (config or {}).get("config", ""), "seed": seed},
                      "metrics": {"train": last_train, "valid": best_valid, "test": out.get("test", {}), "best": out.get("test", {})}}],
            "timestamp": now.strftime("%Y-%m-%dT%H:%M:%S"), "config": config or {},
        }]

    def save_results(self, path, dataset, model_name, out, **kw):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "results.json"), "w") as f:
            json.dump(self.results_record(dataset, model_name, out, **kw), f, indent=2)

    def evaluate(self, mode="valid"):
        pipe = self.validpipe if mode == "valid" else self.testpipe
        if self.kind == "pred":
            ev = PredictionEvaluator(self.monitors)
            was_training = self.model.training
            self.model.eval()
            for data in pipe:
                data = self.dict_to_device(data)
                ev.update(self.model.encode(data["X"])[0], data["Label"])
            self.model.train(was_training)
            return ev.compute()
        ev = RankingEvaluator(self.monitors)
        was_training = getattr(self.model, "training", False)
        if hasattr(self.model, "eval"):
            self.model.eval()
        if hasattr(self.model, "reset_ranking_buffers"):
            self.model.reset_ranking_buffers()
        for data in pipe:
            seen_ptr, seen_idx = ragged_to_csr(data["ISeen"], self.device)
            tgt_ptr, tgt_idx = ragged_to_csr(data["IUnseen"], self.device)
            if self.kind == "module":
                _, idx = self.model.recommend_topk(data[self.fit_keys[0]].to(self.device), seen_ptr, seen_idx, ev.kmax)
            elif self.kind == "seq":
                _, idx = self.model.recommend_topk(data["ISeq"].to(self.device), seen_ptr, seen_idx, ev.kmax)
            else:
                _, idx = self.model.recommend_topk(data["User"].to(self.device), seen_ptr, seen_idx, ev.kmax)
            ev.update(idx, tgt_ptr, tgt_idx)
        if hasattr(self.model, "train"):
            self.model.train(was_training)
        return ev.compute()

    #: monitors for which smaller is better (freerec's DEFAULT_BEST_CASTER: min for losses, max for ranking / AUC metrics)
    MINIMISED = ("LOSS", "LOGLOSS", "MSE", "MAE", "RMSE")

    @property
    def best_mode(self):
        return "min" if self.which4best.split("@")[0].upper() in self.MINIMISED else "max"

    def _better(self, score, best):
        return score < best if self.best_mode == "min" else score > best

    def fit(self, epochs):
        if self.lr_scheduler is not None and getattr(self.lr_scheduler, "mode", self.best_mode) != self.best_mode:
            import warnings
            warnings.warn(f"lr_scheduler.mode={self.lr_scheduler.mode!r} but which4best={self.which4best!r} is {self.best_mode}imised: "
                          "the scheduler will read every improvement as a bad epoch")
        for epoch in range(1, epochs + 1):
            rec = {"epoch": epoch, "train": self.train_per_epoch(epoch)}
            if self.validpipe is not None and epoch % self.eval_freq == 0:
                rec["valid"] = self.evaluate("valid")
                if "@" in self.which4best:
                    name, k = self.which4best.split("@")
                    score = rec["valid"].get(f"{name.upper()}@{k}")
                else:
                    score = rec["valid"].get(self.which4best.upper())
                if score is not None and (self.best is None or self._better(score, self.best[1])):
                    self.best = (epoch, score)
                    if self.checkpoint_path:
                        self.save_best(self.checkpoint_path)
            self.history.append(rec)
            if self.checkpoint_path:
                self.save_checkpoint(self.checkpoint_path, epoch)
        out = {"history": self.history, "best": self.best}
        if self.testpipe is not None:
            out["test"] = self.evaluate("test")
        return out
