"""Full-ranking evaluation on the engine: the host-side mirror of `freerec.launcher.Coach.evaluate` (external to the
reference; contract mirrored at UniSRec/main.py:400-447) for dot-product models.

Reference per batch: dense scores [B,N] -> `scores[seen] = -1e23` -> dense targets [B,N] -> one metric function per
"NAME@K" in cfg.monitors -> `monitor(..., n=bsz, reduction="mean")`.  Here: one fused score+mask+top-K launch, one
metrics launch, running sums on the device; a single host read at the end of the split.
"""
import torch

from . import ops


def parse_monitors(monitors):
    """["LOSS", "HitRate@10", "NDCG@10", ...] -> sorted list of (NAME, k); names upper-cased as freerec does
    (UniSRec/main.py:360-365).  Entries without "@" (LOSS) are not ranking metrics."""
    out = []
    for m in monitors:
        if "@" in m:
            name, k = m.split("@")
            out.append((name.upper(), int(k)))
    return out


def ragged_to_csr(lists, device, sort=True):
    """list of per-user id lists -> (ptr int64[B+1], idx int64[nnz]) on `device`; ids ascending per user."""
    ptr = torch.zeros(len(lists) + 1, dtype=torch.int64)
    ptr[1:] = torch.cumsum(torch.tensor([len(x) for x in lists], dtype=torch.int64), 0)
    flat = [torch.as_tensor(sorted(x) if sort else list(x), dtype=torch.int64) for x in lists]
    idx = torch.cat(flat) if flat else torch.zeros(0, dtype=torch.int64)
    return ptr.to(device), idx.to(device)


class RankingEvaluator:
    """Accumulates HITRATE/PRECISION/RECALL/NDCG/MRR @ k over the batches of one split."""

    def __init__(self, monitors, kmax=None):
        self.wanted = parse_monitors(monitors)
        self.ks = sorted({k for _, k in self.wanted}) or [10]
        self.kmax = kmax or max(self.ks)
        self.sums = None
        self.n = 0

    def update(self, topk_idx, tgt_ptr, tgt_idx):
        _, s = ops.rank_metrics(topk_idx, tgt_ptr, tgt_idx, self.ks)
        self.sums = s if self.sums is None else self.sums + s
        self.n += topk_idx.shape[0]

    def evaluate_batch(self, model_topk, tgt_ptr, tgt_idx):
        """model_topk() -> (vals, idx) from `recommend_topk`; convenience wrapper."""
        _, idx = model_topk(self.kmax)
        self.update(idx, tgt_ptr, tgt_idx)

    def compute(self):
        """-> {"HITRATE@10": float, ...} for the monitors asked for (mean over users)."""
        if self.sums is None:                       # an empty split: every metric is 0
            return {f"{name}@{k}": 0.0 for name, k in self.wanted}
        s = (self.sums / max(self.n, 1)).cpu()
        out = {}
        for name, k in self.wanted:
            j = ops.METRIC_NAMES.index(name)
            out[f"{name}@{k}"] = float(s[self.ks.index(k), j])
        return out


class PredictionEvaluator:
    """LOGLOSS / AUC over the batches of one split of a prediction model (DeepFM/configs/Frappe_x1_BARS.yaml:101-102): the
    reference's Coach collects `recommend_from_pool` outputs (sigmoid(logits), DeepFM/main.py:217-219) against the labels; here the
    logits of all batches are kept on the device and the two metrics come from one launch each at the end of the split:
    LOGLOSS = mean binary cross entropy (re_bce_logits: the stable logits form of -[y log p + (1 - y) log(1 - p)]), AUC = the
    Mann-Whitney statistic counted pairwise (re_auc: exact, sort-free)."""

    def __init__(self, monitors):
        self.wanted = [m.upper() for m in monitors if m.upper() in ("LOGLOSS", "AUC")]
        self.logits, self.labels = [], []

    def update(self, logits, labels):
        self.logits.append(logits.reshape(-1).to(torch.float32))
        self.labels.append(labels.reshape(-1).to(torch.float32))

    def compute(self):
        if not self.logits:
            return {m: 0.0 for m in self.wanted}
        z, y = torch.cat(self.logits).contiguous(), torch.cat(self.labels).contiguous()
        out = {}
        if "LOGLOSS" in self.wanted:
            out["LOGLOSS"] = float(ops.bce_logits(z, y)[0])
        if "AUC" in self.wanted:
            out["AUC"] = float(ops.auc(torch.sigmoid(z).contiguous(), y))       # (scores as the reference hands them over: probabilities)
        return out


class ReduceLROnPlateau:
    """torch.optim.lr_scheduler.ReduceLROnPlateau as CoachForDeepFM uses it (DeepFM/main.py:251-257: mode="max", patience=eval_freq,
    min_lr / factor / threshold from the yaml, threshold_mode "rel", cooldown 0; `step(self._best)` at the top of every epoch):
    the same state machine on the engine's own learning-rate attribute."""

    def __init__(self, model, mode="max", factor=0.1, patience=10, threshold=1e-4, min_lr=0.0, eps=1e-8):
        assert mode in ("max", "min")
        self.model, self.mode, self.factor, self.patience, self.threshold, self.min_lr, self.eps = model, mode, factor, patience, threshold, min_lr, eps
        self.best = -float("inf") if mode == "max" else float("inf")
        self.num_bad_epochs, self.last_epoch = 0, 0

    def _better(self, a):
        if self.mode == "max":
            return a > self.best * (1.0 + self.threshold)
        return a < self.best * (1.0 - self.threshold)

    def step(self, metric):
        cur = float(metric)
        self.last_epoch += 1
        if self._better(cur):
            self.best, self.num_bad_epochs = cur, 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            new = max(self.model.lr * self.factor, self.min_lr)
            if self.model.lr - new > self.eps:
                self.model.lr = new
            self.num_bad_epochs = 0

    def get_last_lr(self):
        return [self.model.lr]

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "model"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)
