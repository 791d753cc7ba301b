"""The engine's operators as registered PyTorch custom ops: `torch.ops.recengine.*` (SURVEY.md §8b "Torch-op surface").

    recengine::gather_rows(W, idx) -> Tensor                               W[idx]            (nn.Embedding.__call__, MF-BPR/main.py:84-86)
    recengine::scatter_add_rows(g, idx, R, padding_idx) -> Tensor          dense [R, D] gradient of gather_rows (embedding_dense_backward)
    recengine::bpr_triplet(U, I, users, pos, neg) -> (loss, logits)        MF.fit's gathers + row dots + BPRLoss (MF-BPR/main.py:81-93)
    recengine::score_dense(Q, E) -> Tensor                                 einsum("BD,ND->BN") (SASRec/main.py:228)
    recengine::score_topk(Q, E, seen_ptr, seen_idx, K) -> (vals, idx)      Coach.evaluate's masked top-K (UniSRec/main.py:408-414)
    recengine::spmm_csr(crow, col, val, X) -> Tensor                       Adj @ X (LightGCN/main.py:80-84).  The registered backward is
                                                                           A dY, i.e. it is right for SYMMETRIC A only (the reference's
                                                                           to_normalized_adj("sym")); for any other matrix use
                                                                           recboard_amd.nn.spmm(A, At, X), which takes the transpose explicitly

Each op has a HIP implementation (ctypes -> librecengine.so on the current stream; there is no CPU kernel and CPU tensors raise),
a fake (meta) implementation so that it traces / exports, and -- where the reference differentiates through it -- a registered
autograd formula whose backward runs on the engine's kernels too.  `recboard_amd.nn` is the module-level surface over these ops.
"""
import weakref

import torch

from . import ops

_lib = torch.library


@_lib.custom_op("recengine::gather_rows", mutates_args=(), device_types="cuda")
def gather_rows(W: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    return ops.gather_rows(W.contiguous(), idx.contiguous())


@gather_rows.register_fake
def _(W, idx):
    return W.new_empty(tuple(idx.shape) + (W.shape[1],))


@_lib.custom_op("recengine::scatter_add_rows", mutates_args=(), device_types="cuda")
def scatter_add_rows(g: torch.Tensor, idx: torch.Tensor, R: int, padding_idx: int = -1) -> torch.Tensor:
    return ops.scatter_add_rows(g.contiguous().view(-1, g.shape[-1]), idx.contiguous().view(-1), R, padding_idx)


@scatter_add_rows.register_fake
def _(g, idx, R, padding_idx=-1):
    return g.new_empty((R, g.shape[-1]))


def _gather_setup(ctx, inputs, output):
    W, idx = inputs
    ctx.save_for_backward(idx)
    ctx.rows = W.shape[0]


def _gather_bwd(ctx, g):
    (idx,) = ctx.saved_tensors
    return torch.ops.recengine.scatter_add_rows(g, idx, ctx.rows, -1), None


gather_rows.register_autograd(_gather_bwd, setup_context=_gather_setup)


def _scatter_setup(ctx, inputs, output):
    g, idx, R, padding_idx = inputs
    ctx.save_for_backward(idx)
    ctx.padding_idx, ctx.gshape = padding_idx, g.shape


def _scatter_bwd(ctx, dW):
    (idx,) = ctx.saved_tensors
    dg = torch.ops.recengine.gather_rows(dW, idx.reshape(-1))
    if ctx.padding_idx >= 0:
        dg = dg * (idx.reshape(-1, 1) != ctx.padding_idx)
    return dg.view(ctx.gshape), None, None, None


scatter_add_rows.register_autograd(_scatter_bwd, setup_context=_scatter_setup)


@_lib.custom_op("recengine::bpr_triplet", mutates_args=(), device_types="cuda")
def bpr_triplet(U: torch.Tensor, I: torch.Tensor, users: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    users, pos, neg = (t.reshape(-1).contiguous() for t in (users, pos, neg))
    loss, logits = ops.bpr_triplet_fwd(U.contiguous(), I.contiguous(), users, pos, neg)
    return loss.squeeze(0), logits


@bpr_triplet.register_fake
def _(U, I, users, pos, neg):
    return U.new_empty(()), U.new_empty((users.numel(), 2))


@_lib.custom_op("recengine::bpr_triplet_backward", mutates_args=(), device_types="cuda")
def bpr_triplet_backward(U: torch.Tensor, I: torch.Tensor, users: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, logits: torch.Tensor,
                         dloss: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    users, pos, neg = (t.reshape(-1).contiguous() for t in (users, pos, neg))
    gu, gp, gn = ops.bpr_triplet_bwd(U.contiguous(), I.contiguous(), users, pos, neg, logits, dloss.reshape(1).contiguous())
    dU = ops.scatter_add_rows(gu, users, U.shape[0])
    dI = ops.scatter_add_rows(torch.cat([gp, gn]), torch.cat([pos, neg]), I.shape[0])
    return dU, dI


@bpr_triplet_backward.register_fake
def _(U, I, users, pos, neg, logits, dloss):
    return torch.empty_like(U), torch.empty_like(I)


def _bpr_setup(ctx, inputs, output):
    U, I, users, pos, neg = inputs
    ctx.save_for_backward(U, I, users, pos, neg, output[1])


def _bpr_bwd(ctx, dloss, dlogits):
    U, I, users, pos, neg, logits = ctx.saved_tensors
    dU, dI = torch.ops.recengine.bpr_triplet_backward(U, I, users, pos, neg, logits, dloss)
    return dU, dI, None, None, None


bpr_triplet.register_autograd(_bpr_bwd, setup_context=_bpr_setup)


@_lib.custom_op("recengine::score_dense", mutates_args=(), device_types="cuda")
def score_dense(Q: torch.Tensor, E: torch.Tensor) -> torch.Tensor:
    return ops.score_dense(Q.contiguous(), E.contiguous())


@score_dense.register_fake
def _(Q, E):
    return Q.new_empty((Q.shape[0], E.shape[0]))


@_lib.custom_op("recengine::gemm", mutates_args=(), device_types="cuda")
def gemm(A: torch.Tensor, B: torch.Tensor, transA: bool = False, transB: bool = False) -> torch.Tensor:
    return ops.gemm(A, B, transA=transA, transB=transB)


@gemm.register_fake
def _(A, B, transA=False, transB=False):
    M = A.shape[1] if transA else A.shape[0]
    N = B.shape[0] if transB else B.shape[1]
    return A.new_empty((M, N))


def _gemm_setup(ctx, inputs, output):
    A, B, transA, transB = inputs
    ctx.save_for_backward(A, B)
    ctx.tA, ctx.tB = bool(transA), bool(transB)


def _gemm_bwd(ctx, dC):
    """C = op(A) op(B):  d op(A) = dC op(B)^T,  d op(B) = op(A)^T dC -- each again ONE recengine::gemm (no transposed copies)."""
    A, B = ctx.saved_tensors
    dC = dC.contiguous()
    G = torch.ops.recengine.gemm
    dA = dB = None
    if ctx.needs_input_grad[0]:
        dA = G(B, dC, ctx.tB, True) if ctx.tA else G(dC, B, False, not ctx.tB)
    if ctx.needs_input_grad[1]:
        dB = G(dC, A, True, ctx.tA) if ctx.tB else G(A, dC, not ctx.tA, False)
    return dA, dB, None, None


gemm.register_autograd(_gemm_bwd, setup_context=_gemm_setup)


def _score_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs)


def _score_bwd(ctx, dS):
    Q, E = ctx.saved_tensors
    dS = dS.contiguous()
    dQ = torch.ops.recengine.gemm(dS, E, False, False) if ctx.needs_input_grad[0] else None      # [B,N] @ [N,D]
    dE = torch.ops.recengine.gemm(dS, Q, True, False) if ctx.needs_input_grad[1] else None       # [N,B] @ [B,D]
    return dQ, dE


score_dense.register_autograd(_score_bwd, setup_context=_score_setup)


@_lib.custom_op("recengine::score_topk", mutates_args=(), device_types="cuda")
def score_topk(Q: torch.Tensor, E: torch.Tensor, seen_ptr: torch.Tensor, seen_idx: torch.Tensor, K: int) -> tuple[torch.Tensor, torch.Tensor]:
    return ops.score_topk(Q.contiguous(), E.contiguous(), seen_ptr, seen_idx, K)


@score_topk.register_fake
def _(Q, E, seen_ptr, seen_idx, K):
    return Q.new_empty((Q.shape[0], K)), Q.new_empty((Q.shape[0], K), dtype=torch.int64)


_PLANS = {}


def _plan(crow, D):
    """The row-order / long-row-chunk plan of an adjacency, built once per `crow` TENSOR.  The cache key is the address, but an entry
    is only valid for the tensor object it was built from at the version it had then: an adjacency rebuilt per epoch (SGL's edge
    dropout as the reference does it, SGL/main.py) usually gets the freed `crow`'s address back from the caching allocator, and an
    in-place edit keeps it -- a stale plan would then read col / val past nnz."""
    key = (crow.data_ptr(), crow.numel(), D)
    ent = _PLANS.get(key)
    if ent is not None and ent[1]() is crow and ent[2] == crow._version:
        return ent[0]
    if len(_PLANS) > 16:
        _PLANS.clear()
    plan = ops.spmm_plan(crow, D)
    _PLANS[key] = (plan, weakref.ref(crow), crow._version)
    return plan


@_lib.custom_op("recengine::spmm_csr", mutates_args=(), device_types="cuda")
def spmm_csr(crow: torch.Tensor, col: torch.Tensor, val: torch.Tensor, X: torch.Tensor) -> torch.Tensor:
    X = X.contiguous()
    return ops.spmm_csr(crow, col, val, _plan(crow, X.shape[1]), X, torch.empty_like(X))


@spmm_csr.register_fake
def _(crow, col, val, X):
    return X.new_empty((crow.numel() - 1, X.shape[1]))


def _spmm_setup(ctx, inputs, output):
    crow, col, val, X = inputs
    ctx.save_for_backward(crow, col, val)


def _spmm_bwd(ctx, dY):
    crow, col, val = ctx.saved_tensors
    # the reference's adjacency is symmetric (LightGCN/main.py:47-49: to_normalized_adj("sym") of an undirected bipartite graph):
    # A^T dY = A dY, the same kernel
    return None, None, None, torch.ops.recengine.spmm_csr(crow, col, val, dY.contiguous())


spmm_csr.register_autograd(_spmm_bwd, setup_context=_spmm_setup)


def _no_cpu(name):
    def raiser(*a, **k):
        raise RuntimeError(f"recengine::{name}: tensors must be on a HIP device (no CPU fallback exists)")
    return raiser


for _name, _op in (("gather_rows", gather_rows), ("scatter_add_rows", scatter_add_rows), ("bpr_triplet", bpr_triplet),
                   ("bpr_triplet_backward", bpr_triplet_backward), ("score_dense", score_dense), ("gemm", gemm), ("score_topk", score_topk),
                   ("spmm_csr", spmm_csr)):
    _op.register_kernel("cpu")(_no_cpu(_name))      # a CPU call fails loudly, with the engine's own message
