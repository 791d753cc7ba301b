"""Oracle: MF-BPR fit / full scores (TEST INFRASTRUCTURE).  torch-CPU fp32.

  fit                  MF-BPR/main.py:81-93   gather u [B,1,D], i+ [B,1,D], i- [B,K,D]; row dots; BPRLoss(mean)
  recommend_from_full  MF-BPR/main.py:101-104 scores = U[users] . I^T  (einsum "BKD,ND->BN", K = 1)
"""
import torch

from . import criterions


def fit(U, I, users, pos, neg):
    u, ip, ineg = U[users], I[pos], I[neg]            # [B,1,D], [B,1,D], [B,K,D]
    return criterions.bpr_loss((u * ip).sum(-1), (u * ineg).sum(-1))


def recommend_from_full(U, I, users):
    return torch.einsum("bkd,nd->bn", U[users], I)
