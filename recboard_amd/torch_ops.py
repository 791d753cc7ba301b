"""The engine's operators as registered PyTorch custom ops: `torch.ops.recengine.*` (SURVEY.md §8b "Torch-op surface").

    recengine::gather_rows(W, idx) -> Tensor                               W[idx]            (nn.Embedding.__call__, MF-BPR/main.py:84-86)
    recengine::scatter_add_rows(g, idx, R, padding_idx) -> Tensor          dense [R, D] gradient of gather_rows (embedding_dense_backward)
    recengine::bpr_triplet(U, I, users, pos, neg) -> (loss, logits)        MF.fit's gathers + row dots + BPRLoss (MF-BPR/main.py:81-93)
    recengine::score_dense(Q, E) -> Tensor                                 einsum("BD,ND->BN") (SASRec/main.py:228)
    recengine::score_topk(Q, E, seen_ptr, seen_idx, K) -> (vals, idx)      Coach.evaluate's masked top-K (UniSRec/main.py:408-414)
    recengine::sasrec_encoder(E, P, seq, params, scale, drop_p, seed) -> (u, tape, plan)
                                                                           SASRec.encode: embedding front end + every block + lastLN fused
                                                                           (SASRec/main.py:163-193, :31-50); backward = re_sasrec_encoder_bwd
    recengine::bce_pair(U, E, pos, neg, valid, kind) -> (loss, logits)     the pair criteria on the rows of U (SASRec/main.py:205-215)
    recengine::fm_bag(T, TL, lr_bias, offsets, x) -> (E, fm_lr)            multi-field bag + FM + LR (DeepFM/main.py:58-62,80-85,204-206)
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


# ---- SASRec.encode as ONE op: embedding front end (E[seq] sqrt(D) + P, dropout, pad mask), every block, lastLN (SASRec/main.py:178-193)
#      params: the 12 * L block tensors in re_sasrec_encoder_fwd's order, then lastLN.weight, lastLN.bias.  tape / plan are what the
#      backward needs (opaque); u rows at pad positions in front of a sequence are not written (nothing on the path reads them).
@_lib.custom_op("recengine::sasrec_encoder", mutates_args=(), device_types="cuda")
def sasrec_encoder(E: torch.Tensor, P: torch.Tensor, seq: torch.Tensor, params: list[torch.Tensor], scale: float, drop_p: float,
                   seed: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    L = (len(params) - 2) // 12
    plan = ops.sasrec_plan(seq.contiguous(), E.shape[1]).clone()      # (a tensor of its own: custom-op outputs may not be views)
    B, S = seq.shape
    u = torch.zeros((B, S, E.shape[1]), dtype=torch.float32, device=E.device)
    _, tape = ops.sasrec_embed_encoder_fwd(E.contiguous(), P.contiguous(), seq.contiguous(), scale, [t.contiguous() for t in params[:-2]],
                                           params[-2].contiguous(), params[-1].contiguous(), L, drop_p, seed, need_tape=True, out=u, plan=plan)
    return u, tape, plan


@sasrec_encoder.register_fake
def _(E, P, seq, params, scale, drop_p, seed):
    from . import lib
    B, S = seq.shape
    D, L = E.shape[1], (len(params) - 2) // 12
    Lb = lib.load()                                        # (pure host size queries)
    return (E.new_empty((B, S, D)), E.new_empty((int(Lb.re_sasrec_tape_bytes(B, S, D, L)) // 4,)),
            seq.new_empty((int(Lb.re_sasrec_plan_bytes(B, S)),), dtype=torch.uint8))


@_lib.custom_op("recengine::sasrec_encoder_backward", mutates_args=(), device_types="cuda")
def sasrec_encoder_backward(dU: torch.Tensor, E: torch.Tensor, P: torch.Tensor, seq: torch.Tensor, params: list[torch.Tensor], scale: float,
                            drop_p: float, seed: int, tape: torch.Tensor, plan: torch.Tensor) -> list[torch.Tensor]:
    L = (len(params) - 2) // 12
    grads = [torch.zeros_like(t) for t in params]
    dP = torch.zeros_like(P)
    contrib = ops.sasrec_encoder_bwd(dU.contiguous(), seq.contiguous(), [t.contiguous() for t in params[:-2]], params[-2].contiguous(),
                                     params[-1].contiguous(), L, drop_p, seed, tape, grads[:-2], grads[-2], grads[-1], plan=plan,
                                     embed_scale=scale, dP=dP)
    dE = ops.scatter_add_rows(contrib.view(-1, E.shape[1]), seq.reshape(-1), E.shape[0], 0)      # (row 0 = the padding row: no gradient)
    return [dE, dP] + grads


@sasrec_encoder_backward.register_fake
def _(dU, E, P, seq, params, scale, drop_p, seed, tape, plan):
    return [torch.empty_like(E), torch.empty_like(P)] + [torch.empty_like(t) for t in params]


def _enc_setup(ctx, inputs, output):
    E, P, seq, params, scale, drop_p, seed = inputs
    ctx.save_for_backward(E, P, seq, output[1], output[2], *params)
    ctx.scale, ctx.drop_p, ctx.seed = scale, drop_p, seed


def _enc_bwd(ctx, dU, dtape, dplan):
    E, P, seq, tape, plan, *params = ctx.saved_tensors
    g = torch.ops.recengine.sasrec_encoder_backward(dU, E, P, seq, list(params), ctx.scale, ctx.drop_p, ctx.seed, tape, plan)
    return g[0], g[1], None, g[2:], None, None, None


sasrec_encoder.register_autograd(_enc_bwd, setup_context=_enc_setup)


# ---- the pair criteria on rows of U: mean over the valid rows of BCE(<u, E[pos]>, 1) + BCE(<u, E[neg]>, 0) (kind 0) or of
#      softplus(<u, E[neg]> - <u, E[pos]>) (kind 1: BPR) -- gathers, dots and criterion in one kernel each way (SASRec/main.py:205-215)
@_lib.custom_op("recengine::bce_pair", mutates_args=(), device_types="cuda")
def bce_pair(U: torch.Tensor, E: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, valid: torch.Tensor, kind: int = 0,
             e_off: int = 0) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    loss, logits, count = ops.pair_loss_fwd(U.contiguous(), E.contiguous(), pos.reshape(-1).contiguous(), neg.reshape(-1).contiguous(),
                                            valid.reshape(-1).to(torch.uint8).contiguous(), kind, e_off)
    return loss.squeeze(0), logits, count


@bce_pair.register_fake
def _(U, E, pos, neg, valid, kind=0, e_off=0):
    return U.new_empty(()), U.new_empty((U.shape[0], 2)), U.new_empty((1,), dtype=torch.int32)


@_lib.custom_op("recengine::bce_pair_backward", mutates_args=(), device_types="cuda")
def bce_pair_backward(U: torch.Tensor, E: torch.Tensor, pos: torch.Tensor, neg: torch.Tensor, valid: torch.Tensor, kind: int, e_off: int,
                      logits: torch.Tensor, count: torch.Tensor, dloss: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    pos, neg = pos.reshape(-1).contiguous(), neg.reshape(-1).contiguous()
    v8 = valid.reshape(-1).to(torch.uint8).contiguous()
    dU, gp, gn = ops.pair_loss_bwd(U.contiguous(), E.contiguous(), pos, neg, v8, kind, logits, count, dloss.reshape(1).contiguous(), e_off)
    live = v8.bool()
    rows = torch.cat([torch.where(live, pos + e_off, torch.full_like(pos, -1)), torch.where(live, neg + e_off, torch.full_like(neg, -1))])
    dE = ops.scatter_add_rows(torch.cat([gp, gn]), rows, E.shape[0], -1)
    return dU, dE


@bce_pair_backward.register_fake
def _(U, E, pos, neg, valid, kind, e_off, logits, count, dloss):
    return torch.empty_like(U), torch.empty_like(E)


def _pair_setup(ctx, inputs, output):
    U, E, pos, neg, valid, kind, e_off = inputs
    ctx.save_for_backward(U, E, pos, neg, valid, output[1], output[2])
    ctx.kind, ctx.e_off = kind, e_off


def _pair_bwd(ctx, dloss, dlogits, dcount):
    U, E, pos, neg, valid, logits, count = ctx.saved_tensors
    dU, dE = torch.ops.recengine.bce_pair_backward(U, E, pos, neg, valid, ctx.kind, ctx.e_off, logits, count, dloss)
    return dU, dE, None, None, None, None, None


bce_pair.register_autograd(_pair_bwd, setup_context=_pair_setup)


# ---- DeepFM's front end: every field's embedding row, the FM second-order term and the logistic-regression term of a row in one
#      kernel (DeepFM/main.py:58-62 LogisticRegression, :80-85 InnerProductInteraction, :204-206): the F field tables are ONE table T
#      [R, D] (+ TL [R]: the LR weights), field f's rows start at offsets[f].  E [B, F * D] feeds the MLP; fm_lr [B] = lr + fm.
@_lib.custom_op("recengine::fm_bag", mutates_args=(), device_types="cuda")
def fm_bag(T: torch.Tensor, TL: torch.Tensor, lr_bias: torch.Tensor, offsets: torch.Tensor, x: torch.Tensor) -> tuple[torch.Tensor, torch.Tensor]:
    return ops.fm_bag_fwd(T.contiguous(), TL.reshape(-1).contiguous(), lr_bias.reshape(1).contiguous(), offsets.contiguous(), x.contiguous())


@fm_bag.register_fake
def _(T, TL, lr_bias, offsets, x):
    return T.new_empty((x.shape[0], x.shape[1] * T.shape[1])), T.new_empty((x.shape[0],))


@_lib.custom_op("recengine::fm_bag_backward", mutates_args=(), device_types="cuda")
def fm_bag_backward(E: torch.Tensor, dE: torch.Tensor, dfm: torch.Tensor, offsets: torch.Tensor, x: torch.Tensor, R: int,
                    D: int) -> tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    F = x.shape[1]
    gE, gL = ops.fm_bag_bwd(E.contiguous(), dE.contiguous(), dfm.contiguous(), F, D)
    rows = (x + offsets.unsqueeze(0)).reshape(-1).contiguous()
    dT = ops.scatter_add_rows(gE, rows, R)
    dTL = ops.scatter_add_rows(gL, rows, R)
    return dT, dTL.reshape(-1), dfm.sum().reshape(1)


@fm_bag_backward.register_fake
def _(E, dE, dfm, offsets, x, R, D):
    return E.new_empty((R, D)), E.new_empty((R,)), E.new_empty((1,))


def _bag_setup(ctx, inputs, output):
    T, TL, lr_bias, offsets, x = inputs
    ctx.save_for_backward(output[0], offsets, x)
    ctx.R, ctx.D, ctx.tl_shape, ctx.b_shape = T.shape[0], T.shape[1], TL.shape, lr_bias.shape


def _bag_bwd(ctx, dE, dfm):
    E, offsets, x = ctx.saved_tensors
    dT, dTL, db = torch.ops.recengine.fm_bag_backward(E, dE, dfm, offsets, x, ctx.R, ctx.D)
    return dT, dTL.view(ctx.tl_shape), db.view(ctx.b_shape), None, None


fm_bag.register_autograd(_bag_bwd, setup_context=_bag_setup)


def _no_cpu(name):
    def raiser(*a, **k):
        raise RuntimeError(f"recengine::{name}: tensors must be on a HIP device (no CPU fallback exists)")
    return raiser


for _name, _op in (("gather_rows", gather_rows), ("scatter_add_rows", scatter_add_rows), ("bpr_triplet", bpr_triplet),
                   ("bpr_triplet_backward", bpr_triplet_backward), ("score_dense", score_dense), ("gemm", gemm), ("score_topk", score_topk),
                   ("spmm_csr", spmm_csr), ("sasrec_encoder", sasrec_encoder), ("sasrec_encoder_backward", sasrec_encoder_backward),
                   ("bce_pair", bce_pair), ("bce_pair_backward", bce_pair_backward), ("fm_bag", fm_bag), ("fm_bag_backward", fm_bag_backward)):
    _op.register_kernel("cpu")(_no_cpu(_name))      # a CPU call fails loudly, with the engine's own message
