"""`freerec.criterions`, restated from the scripts' call sites (parity unpinned: SURVEY.md section 8c):
  BPRLoss(pos, neg)             softplus(neg - pos)           MF-BPR/main.py:44,88-91 (untrained value ln 2), LightGCN/main.py:51,95
  BCELoss4Logits(logits, y)     BCE with logits               SASRec/main.py:121-126,211-214; DeepFM/main.py:168,214
  CrossEntropy4Logits(l, y)     F.cross_entropy               SASRec/main.py:216-219
  BaseCriterion.regularize      sum ||p||^2 / 2 ("l2")        LightGCN/main.py:99-106 (mirrors MF.reg_loss, MF-BPR/main.py:70-76)
subclass contract: `reduction` in mean / sum / none kept on `self.reduction` (SimpleX/main.py:62-86)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BaseCriterion(nn.Module):
    def __init__(self, reduction: str = "mean"):
        super().__init__()
        assert reduction in ("none", "sum", "mean"), f"Invalid reduction of {reduction} ..."
        self.reduction = reduction

    def _reduce(self, x):
        return x.mean() if self.reduction == "mean" else x.sum() if self.reduction == "sum" else x

    @staticmethod
    def regularize(params, rtype: str = "l2"):
        params = [params] if isinstance(params, torch.Tensor) else list(params)
        if rtype == "l1":
            return sum(p.abs().sum() for p in params)
        if rtype == "l2":
            return sum(p.pow(2).sum() for p in params) / 2
        raise NotImplementedError(f"{rtype} regularization is not supported ...")


class BPRLoss(BaseCriterion):
    def forward(self, pos_scores, neg_scores):
        return self._reduce(F.softplus(neg_scores - pos_scores))


class BCELoss4Logits(BaseCriterion):
    def forward(self, logits, targets):
        return F.binary_cross_entropy_with_logits(logits, targets.to(logits.dtype), reduction=self.reduction)


class CrossEntropy4Logits(BaseCriterion):
    def forward(self, logits, targets):
        return F.cross_entropy(logits, targets, reduction=self.reduction)


class MSELoss(BaseCriterion):
    def forward(self, inputs, targets):
        return F.mse_loss(inputs, targets.to(inputs.dtype), reduction=self.reduction)


def cross_entropy_with_logits(logits, targets, reduction: str = "mean"):
    return F.cross_entropy(logits, targets, reduction=reduction)


def binary_cross_entropy_with_logits(logits, targets, reduction: str = "mean"):
    return F.binary_cross_entropy_with_logits(logits, targets.to(logits.dtype), reduction=reduction)
