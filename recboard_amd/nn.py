"""Autograd surface of the engine: the hot-path operators as differentiable torch ops and drop-in modules, so that a model
written against torch / FreeRec picks up the HIP kernels without a hand-written step loop.

    reference call site                                   here
    nn.Embedding.__call__, W[idx]       (MF-BPR/main.py:84-86, SASRec/main.py:183)   gather_rows / Embedding
    einsum row dots + BPRLoss           (MF-BPR/main.py:88-91, LightGCN/main.py:95)   bpr_triplet / BPRTripletLoss
    einsum("BD,ND->BN") full scores     (SASRec/main.py:228, MF-BPR/main.py:104)      score_full
    Adj @ X                             (LightGCN/main.py:80-84)                      spmm_sym
    freerec.criterions.{BPRLoss,BCELoss4Logits}  on logits                            BPRLoss / BCELoss4Logits (restated from call sites)

Forward and backward both run through the C ABI (recboard_amd.ops); there is no CPU fallback -- CPU tensors raise.  The engines
in sasrec.py / gen.py / deepfm.py remain the fast path (fused steps, one arena, hipGraph replay); these ops are the drop-in path.
"""
import torch

from . import ops


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, W, idx, padding_idx):
        idx = idx.contiguous()
        ctx.save_for_backward(idx)
        ctx.rows, ctx.padding_idx = W.shape[0], padding_idx
        return ops.gather_rows(W.contiguous(), idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        # dense table gradient like aten::embedding_dense_backward, deterministic (sorted segments, no atomics)
        dW = ops.scatter_add_rows(g.contiguous().view(-1, g.shape[-1]), idx.view(-1), ctx.rows, ctx.padding_idx)
        return dW, None, None


def gather_rows(W, idx, padding_idx=-1):
    """W[idx] with a deterministic dense gradient; rows equal to `padding_idx` receive no gradient (nn.Embedding's rule)."""
    return _GatherRows.apply(W, idx, padding_idx)


class Embedding(torch.nn.Module):
    """Drop-in for torch.nn.Embedding(num_embeddings, embedding_dim, padding_idx=None) on the engine's gather / scatter-add."""

    def __init__(self, num_embeddings, embedding_dim, padding_idx=None, device=None):
        super().__init__()
        self.num_embeddings, self.embedding_dim, self.padding_idx = num_embeddings, embedding_dim, padding_idx
        self.weight = torch.nn.Parameter(torch.empty(num_embeddings, embedding_dim, device=device))
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.normal_(self.weight)
        if self.padding_idx is not None:
            with torch.no_grad():
                self.weight[self.padding_idx].zero_()

    def forward(self, idx):
        return gather_rows(self.weight, idx, -1 if self.padding_idx is None else self.padding_idx)


class _BprTriplet(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Ut, It, users, pos, neg):
        users, pos, neg = (t.reshape(-1).contiguous() for t in (users, pos, neg))
        Ut, It = Ut.contiguous(), It.contiguous()
        loss, logits = ops.bpr_triplet_fwd(Ut, It, users, pos, neg)
        ctx.save_for_backward(Ut, It, users, pos, neg, logits)
        return loss.squeeze(0)

    @staticmethod
    def backward(ctx, dloss):
        Ut, It, users, pos, neg, logits = ctx.saved_tensors
        gu, gp, gn = ops.bpr_triplet_bwd(Ut, It, users, pos, neg, logits, dloss.reshape(1).contiguous())
        dU = ops.scatter_add_rows(gu, users, Ut.shape[0])
        dI = ops.scatter_add_rows(torch.cat([gp, gn]), torch.cat([pos, neg]), It.shape[0])
        return dU, dI, None, None, None


def bpr_triplet(Ut, It, users, pos, neg):
    """mean(softplus(<u, i-> - <u, i+>)) over the triplets, fused (gathers, dots, criterion in one kernel each way):
    MF.fit (MF-BPR/main.py:81-93) for one negative per positive."""
    return _BprTriplet.apply(Ut, It, users, pos, neg)


class BPRTripletLoss(torch.nn.Module):
    def forward(self, Ut, It, users, pos, neg):
        return bpr_triplet(Ut, It, users, pos, neg)


class _ScoreFull(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Q, E):
        Q, E = Q.contiguous(), E.contiguous()
        ctx.save_for_backward(Q, E)
        return ops.score_dense(Q, E)

    @staticmethod
    def backward(ctx, dS):
        Q, E = ctx.saved_tensors
        dS = dS.contiguous()
        dQ = ops.gemm(dS, E) if ctx.needs_input_grad[0] else None                  # [B,N] @ [N,D]
        dE = ops.gemm(dS, Q, transA=True) if ctx.needs_input_grad[1] else None     # [N,B] @ [B,D]
        return dQ, dE


def score_full(Q, E):
    """scores[b, n] = <Q[b], E[n]> as exact k-ordered fp32 chains (`recommend_from_full`); differentiable (CE over the catalog)."""
    return _ScoreFull.apply(Q, E)


class _SpmmSym(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, crow, col, val, plan):
        X = X.contiguous()
        ctx.csr, ctx.plan = (crow, col, val), plan
        return ops.spmm_csr(crow, col, val, plan, X, torch.empty_like(X))

    @staticmethod
    def backward(ctx, dY):
        crow, col, val = ctx.csr
        dY = dY.contiguous()
        return ops.spmm_csr(crow, col, val, ctx.plan, dY, torch.empty_like(dY)), None, None, None, None   # A symmetric: A^T dY = A dY


def spmm_sym(crow, col, val, X, plan=None):
    """A @ X for a SYMMETRIC CSR matrix (LightGCN's normalised bipartite adjacency): the backward is the same kernel.
    plan = ops.spmm_plan(crow, D) (row order / long-row chunks; build once per matrix)."""
    if plan is None:
        plan = ops.spmm_plan(crow, X.shape[1])
    return _SpmmSym.apply(X, crow, col, val, plan)


class BPRLoss(torch.nn.Module):
    """freerec.criterions.BPRLoss restated from its call sites (MF-BPR/main.py:44,88-91): softplus(neg - pos), reduced."""

    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction = reduction

    def forward(self, pos, neg):
        x = torch.nn.functional.softplus(neg - pos)
        return x.mean() if self.reduction == "mean" else x.sum() if self.reduction == "sum" else x


class BCELoss4Logits(torch.nn.Module):
    """freerec.criterions.BCELoss4Logits restated from its call sites (SASRec/main.py:211-214, DeepFM/main.py:168,214)."""

    def __init__(self, reduction="mean"):
        super().__init__()
        self.reduction = reduction

    def forward(self, logits, targets):
        return torch.nn.functional.binary_cross_entropy_with_logits(logits, targets.to(logits.dtype), reduction=self.reduction)
