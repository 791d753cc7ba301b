"""Oracle: criteria (TEST INFRASTRUCTURE).  torch-CPU fp32.

The classes live in the third-party package `freerec` 1.0.1 (not vendored) => PARITY UNPINNED.
Restated from call sites:
  BPRLoss(reduction="mean")(pos, neg)          MF-BPR/main.py:44,88-91; LightGCN/main.py:51,95-98; SASRec/main.py:215
      = mean(softplus(neg - pos)) = mean(-log sigmoid(pos - neg));  untrained value ln 2 (std 1e-4 init, MF-BPR/main.py:55)
  BCELoss4Logits(reduction="mean")(logits, y)  SASRec/main.py:211-214; DeepFM/main.py:168,214
      = binary_cross_entropy_with_logits(logits, y.to(logits.dtype))
  CrossEntropy4Logits(reduction="mean")(logits, labels)   SASRec/main.py:217-219  = F.cross_entropy
  BaseCriterion.regularize(params, rtype="l2") LightGCN/main.py:99-106 = sum ||p||^2 / 2
      (consistent with the hand-written MF.reg_loss, MF-BPR/main.py:70-76)
"""
import torch
import torch.nn.functional as F


def _reduce(x, reduction):
    if reduction == "mean":
        return x.mean()
    if reduction == "sum":
        return x.sum()
    return x


def bpr_loss(pos: torch.Tensor, neg: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    return _reduce(F.softplus(neg - pos), reduction)


def bce_with_logits(logits: torch.Tensor, targets: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    return F.binary_cross_entropy_with_logits(logits, targets.to(logits.dtype), reduction=reduction)


def cross_entropy(logits: torch.Tensor, labels: torch.Tensor, reduction: str = "mean") -> torch.Tensor:
    return F.cross_entropy(logits, labels, reduction=reduction)


def regularize_l2(params) -> torch.Tensor:
    params = [params] if isinstance(params, torch.Tensor) else params
    return sum(p.pow(2).sum() for p in params) / 2
