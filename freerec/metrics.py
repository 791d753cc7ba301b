"""`freerec.metrics` on dense score / target matrices, as `Coach.evaluate` calls them (contract mirrored at UniSRec/main.py:430-445:
`metric(scores, targets, k=k, reduction="none")` per user).  Definitions restated (parity unpinned; the published
benchmark rows satisfy the identities these imply -- tests/golden/benchmark_rows.json): with T = relevant items of a user and the top-k
list by score: HITRATE = [any hit], PRECISION = hits / k, RECALL = hits / |T|, NDCG = DCG / IDCG(min(k, |T|)), MRR = 1 / rank of the
first hit (0 if none).  The engine's fused path (`recboard_amd.evaluate`) computes the same numbers from one sorted top-K list."""
import torch


def _topk_hits(scores, targets, k):
    idx = torch.topk(scores, k, dim=-1).indices
    return torch.gather(targets, -1, idx).to(scores.dtype)


def _reduce(x, reduction):
    return x.mean() if reduction == "mean" else x.sum() if reduction == "sum" else x


def hit_rate(scores, targets, k=10, reduction="mean"):
    return _reduce((_topk_hits(scores, targets, k).sum(-1) > 0).to(scores.dtype), reduction)


def precision(scores, targets, k=10, reduction="mean"):
    return _reduce(_topk_hits(scores, targets, k).sum(-1) / k, reduction)


def recall(scores, targets, k=10, reduction="mean"):
    return _reduce(_topk_hits(scores, targets, k).sum(-1) / targets.sum(-1).clamp_min(1), reduction)


def normalized_dcg(scores, targets, k=10, reduction="mean"):
    hits = _topk_hits(scores, targets, k)
    disc = 1.0 / torch.log2(torch.arange(2, k + 2, device=scores.device, dtype=scores.dtype))
    dcg = (hits * disc).sum(-1)
    nrel = targets.sum(-1).clamp(max=k).long()
    idcg = torch.cumsum(disc, 0)[(nrel - 1).clamp_min(0)]
    return _reduce(torch.where(nrel > 0, dcg / idcg, torch.zeros_like(dcg)), reduction)


def mean_reciprocal_rank(scores, targets, k=10, reduction="mean"):
    hits = _topk_hits(scores, targets, k)
    first = (hits.cumsum(-1) == 1) & (hits > 0)
    rr = (first.to(scores.dtype) / torch.arange(1, k + 1, device=scores.device, dtype=scores.dtype)).sum(-1)
    return _reduce(rr, reduction)


def log_loss(preds, targets, reduction="mean"):
    p = preds.clamp(1e-7, 1 - 1e-7)
    t = targets.to(p.dtype)
    return _reduce(-(t * p.log() + (1 - t) * (1 - p).log()), reduction)


def auroc(preds, targets, reduction="mean"):
    """Mann-Whitney: P(score of a positive > score of a negative), ties one half."""
    t = targets.reshape(-1).bool()
    p = preds.reshape(-1)
    pos, neg = p[t], p[~t]
    if pos.numel() == 0 or neg.numel() == 0:
        return torch.tensor(0.0)
    order = torch.argsort(torch.cat([pos, neg]))
    ranks = torch.empty_like(order, dtype=torch.float64)
    vals = torch.cat([pos, neg])[order]
    ranks[order] = torch.arange(1, order.numel() + 1, dtype=torch.float64, device=order.device)
    # average ranks over ties
    uniq, inv, cnt = torch.unique(vals, return_inverse=True, return_counts=True)
    sums = torch.zeros(uniq.numel(), dtype=torch.float64, device=order.device).index_add_(0, inv, ranks[order])
    ranks[order] = (sums / cnt)[inv]
    u = ranks[: pos.numel()].sum() - pos.numel() * (pos.numel() + 1) / 2
    return (u / (pos.numel() * neg.numel())).to(torch.float32)


DEFAULT_METRICS = {
    "HITRATE": hit_rate, "PRECISION": precision, "RECALL": recall, "NDCG": normalized_dcg, "MRR": mean_reciprocal_rank,
    "LOGLOSS": log_loss, "AUC": auroc,
}
