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
        s = (self.sums / max(self.n, 1)).cpu()
        out = {}
        for name, k in self.wanted:
            j = ops.METRIC_NAMES.index(name)
            out[f"{name}@{k}"] = float(s[self.ks.index(k), j])
        return out
