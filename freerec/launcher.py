"""`freerec.launcher.Coach`: the training / evaluation loop the scripts drive (`Coach(dataset=, trainpipe=, validpipe=, testpipe=, model=,
cfg=).fit()`, SASRec/main.py:278-286).  Written from the call sites: overridable `set_optimizer /
set_lr_scheduler / set_other / train_per_epoch / evaluate`; `self.dataloader`, `dict_to_device`, `monitor(...)`, `register_metric`,
`self._best`, `self.lr_scheduler`, `self.User / Item / ISeq / ... / Size`, `remove_seen`; evaluate contract mirrored at
UniSRec/main.py:400-447; fit loop shape evidenced by ETEGRec/train_etegrec.py:625-650; checkpoint.tar keys by :549-574.

ENGINE ROUTING (cfg.engine = "auto", the default; recboard_amd/bridge.py): a model the recengine has a step for -- SASRec (BCE / BPR / CE),
MF-BPR, LightGCN, DeepFM, recognised by structure -- is trained through the engine instead of the script's `train_per_epoch` ONLY AFTER a
probe has shown, on the first training batch, that the engine's step IS the script's step (same gradients handed to the optimizer, the
update of the torch.optim.Adam the script built, a call pattern the adapter replays); otherwise a warning names the model and the reason
and the script's own torch code runs.  After adoption the module's parameters are views of the engine's arena, so `state_dict()`, the
script's own `recommend_from_full` and checkpoints see the trained values.  Full-ranking evaluation of an adopted model runs on the fused
score + seen-mask + top-K kernel and the metrics kernel (seen / target lists as device CSR built once per split), DeepFM's pool
evaluation on the engine's forward + the LOGLOSS / AUC kernels.  `--engine module` keeps everything on the script's own torch code.

ONE Coach.  `recboard_amd.coach.Coach` (engine objects driven without a script: bench, tests) is a constructor shim over this class: the epoch
loop, evaluation, best tracking, `checkpoint.tar` / `best.pt` and the leaderboard record all live here.  `fit()` ends with `results.json` in
the schema of `benchmark/<dataset>/<model>.json` (benchmark/Amazon2014Beauty_550_LOU/SASRec.json:1-304: a list of `{description, dataset,
tags, runs: [{id, params: {config, seed}, metrics: {train, valid, test, best}}], timestamp, config}`), the file
recboard/scripts/build-data.mjs:95-146 aggregates: a run of an unchanged `main.py` can be dropped into `benchmark/`."""
import datetime
import json
import os
import time
from collections import defaultdict

import torch

from . import ddp, metrics, utils
from .data import tags as T

DEFAULT_METRICS = dict(metrics.DEFAULT_METRICS)
DEFAULT_FMTS = defaultdict(lambda: ".4f", {"LOSS": ".5f"})
DEFAULT_BEST_CASTER = defaultdict(lambda: max, {"LOSS": min, "LOGLOSS": min, "MSE": min, "MAE": min, "RMSE": min})


class EarlyStopError(Exception):
    pass


class _Meter:
    def __init__(self, name, func=None, fmt=".4f", best_caster=max):
        self.name, self.func, self.fmt, self.caster = name, func, fmt, best_caster
        self.history = []
        self.reset()

    def reset(self):
        self.sum, self.n = 0.0, 0

    def update(self, val, n=1, reduction="mean"):
        val = float(val)
        self.sum += val * n if reduction == "mean" else val
        self.n += n

    @property
    def avg(self):
        return self.sum / self.n if self.n else 0.0

    def step(self):
        self.history.append(self.avg if self.n else None)
        self.reset()

    def best(self):
        h = [v for v in self.history if v is not None]
        return self.caster(h) if h else None


class Coach:
    def __init__(self, *, dataset, trainpipe, validpipe, testpipe, model, cfg):
        self.dataset, self.cfg = dataset, cfg
        self.trainpipe, self.validpipe, self.testpipe = trainpipe, validpipe, testpipe
        self.fields = getattr(dataset, "fields", None)
        for name in ("User", "Item", "ISeq", "IPos", "INeg", "IUnseen", "ISeen", "Label", "Size"):
            if hasattr(model, name):
                setattr(self, name, getattr(model, name))
        self.remove_seen = not bool(cfg.get("retain_seen", False))
        self.set_device()
        self.set_model(model)
        self.set_dataloader()
        self.set_optimizer()
        self.set_lr_scheduler()
        self._meters = {m: {} for m in ("train", "valid", "test")}
        self._register_default_monitors()
        self.set_other()
        self._best, self._best_epoch = (float("inf") if self._best_caster() is min else -float("inf")), 0
        self._best_saved = False          # a best.pt written by THIS run (or by the run a checkpoint resumes): only that one is ever loaded back
        self._stopping_steps = 0
        self.history = []                   # one record per epoch: {"epoch", "train", ["valid"], ["test"]}; the last one holds the final evaluations
        self._final = {"train": {}, "valid": {}, "test": {}, "best": {}}
        self.path = cfg.get("checkpoint_path") or os.path.join("logs", str(cfg.get("description", "RecSys")), str(cfg.get("dataset")), str(cfg.get("id") or time.strftime("%m%d%H%M%S")))
        self._engine = self._attach_engine()
        if self._engine is not None and hasattr(self.trainpipe, "to_"):
            # batches sampled on the device where a device sampler exists for the chain; for the fused SASRec step the sampling happens
            # INSIDE the step's batch-preparation launch (tickets instead of tensors)
            self.trainpipe.to_(self.device, fused=self._engine.wants_fused_sampler())

    # ---- set-up hooks the scripts override
    def set_device(self):
        want = str(self.cfg.get("device", "cuda:0"))
        if want.startswith("cuda") and not torch.cuda.is_available():
            want = "cpu"
        if want.startswith("cuda") and ddp.is_distributed():
            want = f"cuda:{ddp.get_local_rank()}"
        self.device = torch.device(want)

    def set_model(self, model):
        self.model = model.to(self.device)

    def get_res_sys_arch(self):
        m = self.model
        return m.module if hasattr(m, "module") and isinstance(m, torch.nn.parallel.DistributedDataParallel) else m

    def set_dataloader(self):
        # (the pipes are batch iterables themselves: batch assembly is vectorised, there are no worker processes to start)
        self.dataloader = self.trainpipe

    def set_optimizer(self):
        cfg, name = self.cfg, str(self.cfg.get("optimizer", "adam")).lower()
        params = self.model.parameters()
        if name == "sgd":
            self.optimizer = torch.optim.SGD(params, lr=cfg.lr, momentum=cfg.get("momentum", 0.9), weight_decay=cfg.weight_decay, nesterov=cfg.get("nesterov", False))
        elif name == "adam":
            self.optimizer = torch.optim.Adam(params, lr=cfg.lr, betas=(cfg.get("beta1", 0.9), cfg.get("beta2", 0.999)), weight_decay=cfg.weight_decay)
        elif name == "adamw":
            self.optimizer = torch.optim.AdamW(params, lr=cfg.lr, betas=(cfg.get("beta1", 0.9), cfg.get("beta2", 0.999)), weight_decay=cfg.weight_decay)
        else:
            raise NotImplementedError(f"Unexpected optimizer {cfg.optimizer} ...")

    def set_lr_scheduler(self):
        self.lr_scheduler = None

    def set_other(self):
        pass

    # ---- monitors
    def _best_caster(self):
        return DEFAULT_BEST_CASTER[str(self.cfg.get("which4best", "LOSS")).split("@")[0].upper()]

    def register_metric(self, name, func=None, fmt=".4f", best_caster=max, prefix=None):
        name = name.upper()
        for mode in self._meters:
            self._meters[mode][name] = _Meter(name, func, fmt, best_caster)

    def _register_default_monitors(self):
        for mon in self.cfg.get("monitors", []) or []:
            fam = mon.split("@")[0].upper()
            self.register_metric(mon.upper(), DEFAULT_METRICS.get(fam), DEFAULT_FMTS[fam], DEFAULT_BEST_CASTER[fam])
        if "LOSS" not in self._meters["train"]:
            self.register_metric("LOSS", None, DEFAULT_FMTS["LOSS"], min)

    def monitor(self, *values, n=1, reduction="mean", mode="train", pool=None):
        """`monitor(loss.item(), n=, reduction=, mode="train", pool=["LOSS"])` or `monitor(scores, targets, n=, mode=, pool=[families])`:
        every registered monitor whose family is in `pool` is updated (a "NAME@k" monitor calls its metric with k)."""
        pool = [p.upper() for p in (pool or [])]
        for name, meter in self._meters[mode].items():
            fam = name.split("@")[0]
            if fam not in pool and name not in pool:
                continue
            if meter.func is None or len(values) == 1 and not torch.is_tensor(values[0]):
                meter.update(values[0], n, reduction)
            else:
                kw = {"k": int(name.split("@")[1])} if "@" in name else {}
                meter.update(meter.func(*values, reduction="mean", **kw), n, "mean")

    def _step_meters(self, mode):
        out = {}
        for name, meter in self._meters[mode].items():
            if meter.n:
                out[name] = meter.avg
            meter.step()
        return out

    # ---- data movement
    def dict_to_device(self, data):
        return {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in data.items()}

    # ---- the loops
    def train_per_epoch(self, epoch):
        raise NotImplementedError("train_per_epoch: the scripts define the step loop (SASRec/main.py:242-258)")

    def _set_training(self, flag):
        m = self.get_res_sys_arch()
        m.train() if flag else m.eval()

    def train(self, epoch):
        self.dataloader = self.trainpipe
        self._set_training(True)
        if self._engine is not None:
            self._engine.train_epoch(self, epoch)
        else:
            self.train_per_epoch(epoch)
        return self._step_meters("train")

    @torch.no_grad()
    def valid(self, epoch, step=-1):
        self.dataloader = self.validpipe
        self._set_training(False)
        self._evaluate(epoch, step, "valid")
        return self._step_meters("valid")

    @torch.no_grad()
    def test(self, epoch, step=-1):
        self.dataloader = self.testpipe
        self._set_training(False)
        self._evaluate(epoch, step, "test")
        return self._step_meters("test")

    def _evaluate(self, epoch, step, mode):
        # a subclass's own `evaluate` wins; otherwise the fused path where it applies, else the reference's dense path
        if type(self).evaluate is not Coach.evaluate or self.cfg.get("engine", "auto") == "module" or not self._fused_eval(mode):
            self.evaluate(epoch, step, mode)

    def evaluate(self, epoch, step=-1, mode="valid"):
        """The reference's dense evaluation (contract: UniSRec/main.py:400-447)."""
        arch = self.get_res_sys_arch()
        arch.reset_ranking_buffers()
        pred = isinstance(arch, __import__("freerec").models.PredRecArch)
        for data in self.dataloader:
            bsz = data.get(self.Size, None) if hasattr(self, "Size") else None
            data = self.dict_to_device(data)
            if pred:
                scores = self.model(data, ranking="pool")
                targets = data[self.Label]
                self.monitor(scores.reshape(-1), targets.reshape(-1), n=len(targets), mode=mode, pool=["LOGLOSS", "AUC"])
                continue
            if self.cfg.ranking == "full":
                scores = self.model(data, ranking="full")
                if self.remove_seen:
                    seen = self.Item.to_csr(data[self.ISeen]).to(self.device).to_dense().bool()
                    scores[seen] = -1e23
                targets = self.Item.to_csr(data[self.IUnseen]).to(self.device).to_dense()
            elif self.cfg.ranking == "pool":
                scores = self.model(data, ranking="pool")
                targets = torch.zeros_like(scores)
                targets[:, 0].fill_(1)
            else:
                raise NotImplementedError(f"`ranking` should be 'full' or 'pool' but {self.cfg.ranking} received ...")
            self.monitor(scores, targets, n=bsz if bsz is not None else scores.shape[0], reduction="mean", mode=mode,
                         pool=["HITRATE", "PRECISION", "RECALL", "NDCG", "MRR"])

    def _fused_eval(self, mode):
        """Full ranking on the engine's fused score + mask + top-K and metrics kernels, for models that expose what a dot-product score
        needs (the engine adapter, or `recommend_topk`).  -> False when it does not apply."""
        if self.device.type != "cuda" or self._engine is None:
            return False
        if hasattr(self._engine, "pool_logits"):              # a prediction model (DeepFM): LOGLOSS / AUC
            covered = ("LOGLOSS", "AUC")
            mons = [m for m in self._meters[mode] if m.split("@")[0] in covered]
            other = [m for m in self._meters[mode] if m not in mons and m != "LOSS"]
            if other or not mons:         # a monitor the kernels do not compute (or none at all): the reference's dense path updates what it can
                return False
            self._engine.reset_ranking_buffers()
            if str(self.cfg.get("pred_metrics", "batch")) == "global":
                # ONE value over all rows of the split (the statistically meaningful AUC; NOT what the reference's Coach monitors)
                from recboard_amd.evaluate import PredictionEvaluator
                ev, n = PredictionEvaluator(mons), 0
                for data in self.dataloader:
                    logits, labels = self._engine.pool_logits(self, data)
                    ev.update(logits, labels)
                    n += int(labels.numel())
                for name, val in ev.compute().items():
                    self._meters[mode][name].update(val, max(n, 1), "mean")
                return True
            # the reference's form (UniSRec/main.py:400-447 with a PredRecArch; DeepFM/main.py:217-219): every batch's LOGLOSS / AUC, weighted by
            # the batch's rows -- the mean of per-batch AUCs, not the split's AUC.  Computed by the device kernels, summed on the device, one host read.
            from recboard_amd import ops
            tot = {m: torch.zeros((), dtype=torch.float64, device=self.device) for m in mons}
            n = 0
            for data in self.dataloader:
                logits, labels = self._engine.pool_logits(self, data)
                z, y = logits.reshape(-1).to(torch.float32).contiguous(), labels.reshape(-1).to(torch.float32).contiguous()
                b = int(y.numel())
                if "LOGLOSS" in tot:
                    tot["LOGLOSS"] += ops.bce_logits(z, y)[0].reshape(()).double() * b
                if "AUC" in tot:
                    tot["AUC"] += ops.auc(torch.sigmoid(z).contiguous(), y).reshape(()).double() * b
                n += b
            for name, t in tot.items():
                self._meters[mode][name].update(float(t) / max(n, 1), max(n, 1), "mean")
            return True
        ranking = self.cfg.get("ranking", "full")
        if ranking not in ("full", "pool") or (ranking == "pool" and not hasattr(self._engine, "recommend_pool")):
            return False
        from recboard_amd import ops
        from recboard_amd.evaluate import RankingEvaluator
        mons = [m for m in self._meters[mode] if "@" in m]
        if not mons:
            return True
        if any(m.split("@")[0] not in ops.METRIC_NAMES for m in mons):     # a ranking monitor the metrics kernel does not have: the dense path
            return False
        if ranking == "pool":
            # `--ranking=pool` (UniSRec/main.py:415-421: scores [B, 1 + K] of the target and K sampled unseen items, targets[:, 0] = 1): the
            # pool's rows gathered and dotted by one launch, the exact top-K of the row (ties to the lowest position -- the target's, as in the
            # full ranking), the metrics kernel against the target list {position 0}
            ev = RankingEvaluator(mons)
            self._engine.reset_ranking_buffers()
            for data in self.dataloader:
                scores = self._engine.recommend_pool(self, data)
                B = scores.shape[0]
                _, idx = ops.pool_topk(scores, ev.kmax)
                ev.update(idx, torch.arange(B + 1, dtype=torch.int64, device=self.device), torch.zeros(B, dtype=torch.int64, device=self.device))
            for name, val in ev.compute().items():
                self._meters[mode][name].update(val, max(ev.n, 1), "mean")
            return True
        ev = RankingEvaluator(mons)
        self._engine.reset_ranking_buffers()
        for j, data in enumerate(self.dataloader):
            seen_ptr, seen_idx, tgt_ptr, tgt_idx = self._split_csr(mode, j, data)
            _, idx = self._engine.recommend_topk(self, data, seen_ptr, seen_idx, ev.kmax)
            ev.update(idx, tgt_ptr, tgt_idx)
        for name, val in ev.compute().items():
            self._meters[mode][name].update(val, max(ev.n, 1), "mean")
        return True

    def _split_csr(self, mode, j, data):
        """The seen / target lists of batch j of a split as device CSR -- built ONCE per split (the evaluation pipes
        are ordered, every pass hands out the same rows) and kept on the device; a batch whose users are not the cached ones is rebuilt."""
        from recboard_amd.evaluate import ragged_to_csr
        cache = self.__dict__.setdefault("_csr_cache", {})
        users = data.get(self.User) if hasattr(self, "User") else None
        sig = users.reshape(-1).cpu() if torch.is_tensor(users) else None
        hit = cache.get((mode, j))
        if hit is not None and sig is not None and hit[0].shape == sig.shape and bool((hit[0] == sig).all()) and hit[1] == self.remove_seen:
            return hit[2]
        empty = [[] for _ in data[self.ISeen]]
        csr = ragged_to_csr(data[self.ISeen] if self.remove_seen else empty, self.device) + ragged_to_csr(data[self.IUnseen], self.device)
        if sig is not None:
            cache[(mode, j)] = (sig, self.remove_seen, csr)
        return csr

    # ---- engine routing
    def _attach_engine(self):
        if self.cfg.get("engine", "auto") == "module" or self.device.type != "cuda":
            return None
        try:
            from recboard_amd import bridge, lib
            lib.load()
        except Exception as e:  # noqa: BLE001  (the engine library is not built / does not load: say so, loudly -- cfg.engine = "auto" asked for it)
            import warnings
            msg = (f"[recengine] >>> librecengine.so is not available ({type(e).__name__}: {e}); the model trains and evaluates on its own torch "
                   "code.  Build it with `python __graft_entry__.py`, or pass --engine module to silence this.")
            warnings.warn(msg)
            utils.warnLogger(msg)
            return None
        return bridge.attach(self)

    # ---- checkpoints / results (checkpoint.tar {epoch, model, optimizer, lr_scheduler, monitors}, best.pt, results.json)
    def _model_state(self):
        return self.get_res_sys_arch().state_dict()

    def _load_model_state(self, sd):
        self.get_res_sys_arch().load_state_dict(sd)

    def save_checkpoint(self, epoch, path=None):
        """`checkpoint.tar` with the reference's keys.  `optimizer` is ALWAYS in torch.optim's state_dict shape over the script's own parameter
        groups (an adopted engine's moments are written per module parameter), so a checkpoint moves between `--engine auto` and
        `--engine module`, and to the reference."""
        path = path or self.path
        utils.mkdirs(path)
        opt = self._engine.optimizer_state(self) if self._engine is not None else (self.optimizer.state_dict() if self.optimizer is not None else {})
        torch.save({"epoch": epoch, "model": self._model_state(), "optimizer": opt,
                    "lr_scheduler": self.lr_scheduler.state_dict() if self.lr_scheduler is not None else None,
                    "monitors": {"meters": {mode: {k: m.history for k, m in ms.items()} for mode, ms in self._meters.items()},
                                 "best": self._best, "best_epoch": self._best_epoch, "history": self.history}},
                   os.path.join(path, "checkpoint.tar"))

    def load_checkpoint(self, path=None):
        ck = torch.load(os.path.join(path or self.path, "checkpoint.tar"), map_location=self.device, weights_only=False)
        self._load_model_state(ck["model"])
        opt = ck.get("optimizer") or {}
        if opt:
            if self._engine is not None:
                # the engine's moments AND the script's optimizer: its param_groups carry the learning rate every epoch starts from
                # (`_Adapter.begin_epoch` reads it there; a ReduceLROnPlateau has lowered it in a resumed DeepFM run)
                self._engine.load_optimizer_state(self, opt)
            elif self.optimizer is not None:
                self.optimizer.load_state_dict(opt)
        if self.lr_scheduler is not None and ck.get("lr_scheduler"):
            self.lr_scheduler.load_state_dict(ck["lr_scheduler"])
        mon = ck.get("monitors") or {}
        meters = mon.get("meters", mon if "best" not in mon else {})       # (checkpoints of round 4: the meters' histories at the top level)
        for mode, ms in meters.items():
            if mode in self._meters and isinstance(ms, dict):
                for k, h in ms.items():
                    if k in self._meters[mode]:
                        self._meters[mode][k].history = list(h)
        if "best" in mon and not isinstance(mon["best"], (tuple, list)):
            self._best, self._best_epoch = mon["best"], mon.get("best_epoch", 0)
        elif isinstance(mon.get("best"), (tuple, list)):                   # (round-4 engine checkpoints: (epoch, score))
            self._best_epoch, self._best = mon["best"]
        self.history = list(mon.get("history", []))
        self._best_saved = self._best not in (float("inf"), -float("inf"))      # (the resumed run's best.pt, if it got as far as writing one)
        return ck["epoch"]

    def save_best(self, path=None):
        path = path or self.path
        utils.mkdirs(path)
        torch.save(self._model_state(), os.path.join(path, "best.pt"))

    def load_best(self, path=None):
        p = os.path.join(path or self.path, "best.pt")
        if os.path.exists(p):
            self._load_model_state(torch.load(p, map_location=self.device))
            return True
        return False

    def resume(self):
        if self.cfg.get("resume") and os.path.exists(os.path.join(self.path, "checkpoint.tar")):
            return self.load_checkpoint() + 1
        return 0

    def check_best(self, epoch, results):
        key = str(self.cfg.get("which4best", "LOSS")).upper()
        if key not in results:
            return
        caster = self._best_caster()
        if caster(results[key], self._best) == results[key] and results[key] != self._best:
            self._best, self._best_epoch, self._stopping_steps = results[key], epoch, 0
            if self._saves_files():
                self.save_best()
                self._best_saved = True
        else:
            self._stopping_steps += 1
            if self._stopping_steps > self.cfg.get("early_stop_patience", 1e23):
                raise EarlyStopError

    def _saves_files(self):
        return True

    def results_record(self, dataset=None, model_name=None, metrics=None, seed=None, config=None, run_id=None, tags=None, description=None):
        """-> the list-of-one record of `benchmark/<dataset>/<model>.json` (benchmark/Amazon2014Beauty_550_LOU/SASRec.json:1-304), what
        recboard/scripts/build-data.mjs:95-146 reads: per run `metrics.{train, valid, test, best}` (train: the last epoch's monitors;
        valid / test: the final model; best: the test split under the best checkpoint -- `aggregateRuns` averages `best`)."""
        cfg = self.cfg
        m = metrics if metrics is not None else self._final
        conf = config if config is not None else {k: v for k, v in cfg.to_dict().items() if isinstance(v, (int, float, str, bool, list, dict, type(None)))}
        now = datetime.datetime.now()
        arch = self.get_res_sys_arch()
        name = model_name or type(arch).__name__
        if tags is None:
            tags = [t for t in (cfg.get("tasktag"), cfg.get("loss"), cfg.get("embedding_dim")) if t is not None]
            tags = [str(t) for t in tags] or [name]
        return [{
            "description": description if description is not None else (str(cfg.get("description") or "") if cfg.get("description") != "RecSys" else ""),
            "dataset": str(dataset if dataset is not None else cfg.get("dataset")),
            "tags": list(tags),
            "runs": [{"id": str(run_id or cfg.get("id") or now.strftime("%m%d%H%M%S")),
                      "params": {"config": conf.get("config") or "", "seed": int(seed if seed is not None else (cfg.get("seed") or 0))},
                      "metrics": {k: {a: float(b) for a, b in (m.get(k) or {}).items()} for k in ("train", "valid", "test", "best")}}],
            "timestamp": now.strftime("%Y-%m-%dT%H:%M:%S"),
            "config": conf,
        }]

    def save_results(self, path=None, **kw):
        path = path or self.path
        utils.mkdirs(path)
        rec = self.results_record(**kw)
        with open(os.path.join(path, "results.json"), "w") as f:
            json.dump(rec, f, indent=2)
        return rec

    def summary(self):
        """Writes `results.json` (the leaderboard record) and returns every monitor's best value per split."""
        best = {mode: {k: m.best() for k, m in ms.items() if m.best() is not None} for mode, ms in self._meters.items()}
        if self._saves_files():
            self.save_results()
        return best

    def _eval_round(self, epoch, rec):
        cfg = self.cfg
        results = {}
        if cfg.get("eval_valid", True) and self.validpipe is not None:
            results = rec["valid"] = self.valid(epoch)
        if cfg.get("eval_test", False) and self.testpipe is not None:
            rec["test"] = self.test(epoch)
        if results:
            self.check_best(epoch, results)

    def fit(self):
        """Per epoch: every `eval_freq` epochs the evaluation round in FRONT of the epoch (valid -> best checkpoint; test if `eval_test`), then
        `train`.  After the last epoch: the final model on valid and test, then the best checkpoint on test -> `results.json`."""
        cfg = self.cfg
        epochs = int(cfg.get("epochs") or 0)
        start = self.resume()
        try:
            for epoch in range(start, epochs):
                rec = {"epoch": epoch}
                if epoch % max(int(cfg.get("eval_freq", 5)), 1) == 0:
                    self._eval_round(epoch, rec)
                rec["train"] = self._final["train"] = self.train(epoch)
                self.history.append(rec)
                utils.infoLogger(f"[Coach] >>> TRAIN @Epoch: {epoch:<4d} >>> " + " || ".join(f"{k} Avg: {v:.5f}" for k, v in rec["train"].items()))
                if cfg.get("checkpoint_freq") and self._saves_files():
                    self.save_checkpoint(epoch)
        except EarlyStopError:
            utils.infoLogger(f"[Coach] >>> Early Stop @Epoch: {epoch}")
        rec = {"epoch": epochs}
        if self.validpipe is not None:
            rec["valid"] = self._final["valid"] = self.valid(epochs)
            try:
                self.check_best(epochs, rec["valid"])
            except EarlyStopError:
                pass
        if self.testpipe is not None:
            rec["test"] = self._final["test"] = self.test(epochs)
        self.history.append(rec)
        if self._saves_files():
            self.save_checkpoint(epochs)
        self._final["best"] = dict(self._final["test"])
        # (only a best.pt of this run: `self.path` is reused between runs, and without a validation split `_best_epoch` stays 0 -- a previous
        #  run's file would otherwise be loaded and its test metrics recorded as this run's `best`)
        if self._saves_files() and self.testpipe is not None and self._best_saved and self._best_epoch != epochs and self.load_best():
            self._final["best"] = self.test(epochs)      # the best checkpoint on the test split: what the leaderboard aggregates
        out = {"valid": self._final["valid"], "test": self._final["test"], "best_test": self._final["best"], "history": self.history,
               "best_epoch": self._best_epoch, "best_value": self._best}
        out["best"] = self.summary()
        return out
