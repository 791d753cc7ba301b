"""A compact Coach for the engine models: host-side mirror of `freerec.launcher.Coach` as the reference scripts use it
(`Coach(dataset=, trainpipe=, validpipe=, testpipe=, model=, cfg=).fit()`, SASRec/main.py:278-286; loop shape evidenced by
ETEGRec/train_etegrec.py:625-650; evaluate contract by UniSRec/main.py:400-447).

Per epoch: `train_per_epoch` (SASRec/main.py:242-258 / MF-BPR/main.py:115-131: one engine `train_step` per batch, LOSS
monitored as the mean over batches weighted by batch size); every `eval_freq` epochs `evaluate("valid")` with the
fused score+mask+top-K kernel and the metrics kernel; best epoch tracked on `which4best`.  The per-step `loss.item()`
host sync of the reference is replaced by one device-side accumulation read at the end of the epoch.
"""
import torch

from .evaluate import RankingEvaluator, ragged_to_csr


class Coach:
    def __init__(self, model, trainpipe, validpipe=None, testpipe=None, monitors=("LOSS", "HitRate@10", "NDCG@10"),
                 which4best="NDCG@10", eval_freq=5, kind="seq"):
        self.model, self.trainpipe, self.validpipe, self.testpipe = model, trainpipe, validpipe, testpipe
        self.monitors, self.which4best, self.eval_freq, self.kind = list(monitors), which4best, eval_freq, kind
        self.history, self.best = [], None
        self.device = model.device

    def dict_to_device(self, data):
        return {k: (v.to(self.device) if isinstance(v, torch.Tensor) else v) for k, v in data.items()}

    def train_per_epoch(self, epoch):
        tot = torch.zeros((), device=self.device)
        n = 0
        for data in self.trainpipe:
            data = self.dict_to_device(data)
            if self.kind == "seq":
                loss = self.model.train_step(data["ISeq"], data["IPos"], data["INeg"])
            else:
                loss = self.model.train_step(data["User"], data["IPos"], data["INeg"])
            bsz = len(data["User"])
            tot += loss * bsz
            n += bsz
        return {"LOSS": float(tot / max(n, 1))}

    def evaluate(self, mode="valid"):
        pipe = self.validpipe if mode == "valid" else self.testpipe
        ev = RankingEvaluator(self.monitors)
        was_training = getattr(self.model, "training", False)
        if hasattr(self.model, "eval"):
            self.model.eval()
        if hasattr(self.model, "reset_ranking_buffers"):
            self.model.reset_ranking_buffers()
        for data in pipe:
            seen_ptr, seen_idx = ragged_to_csr(data["ISeen"], self.device)
            tgt_ptr, tgt_idx = ragged_to_csr(data["IUnseen"], self.device)
            if self.kind == "seq":
                _, idx = self.model.recommend_topk(data["ISeq"].to(self.device), seen_ptr, seen_idx, ev.kmax)
            else:
                _, idx = self.model.recommend_topk(data["User"].to(self.device), seen_ptr, seen_idx, ev.kmax)
            ev.update(idx, tgt_ptr, tgt_idx)
        if hasattr(self.model, "train"):
            self.model.train(was_training)
        return ev.compute()

    def fit(self, epochs):
        for epoch in range(1, epochs + 1):
            rec = {"epoch": epoch, "train": self.train_per_epoch(epoch)}
            if self.validpipe is not None and epoch % self.eval_freq == 0:
                rec["valid"] = self.evaluate("valid")
                name, k = self.which4best.split("@")
                score = rec["valid"].get(f"{name.upper()}@{k}")
                if score is not None and (self.best is None or score > self.best[1]):
                    self.best = (epoch, score)
            self.history.append(rec)
        out = {"history": self.history, "best": self.best}
        if self.testpipe is not None:
            out["test"] = self.evaluate("test")
        return out
