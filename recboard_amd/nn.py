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

from . import torch_ops  # noqa: F401  (registers torch.ops.recengine.*)
from .capture import recording

_R = torch.ops.recengine


def gather_rows(W, idx, padding_idx=-1):
    """W[idx] with a deterministic dense gradient; rows equal to `padding_idx` receive no gradient (nn.Embedding's rule).
    = torch.ops.recengine.gather_rows (+ the padding rule applied to the looked-up rows' gradient)."""
    y = _R.gather_rows(W, idx)
    if padding_idx >= 0:
        # a row looked up at the padding index passes its value on but takes no gradient
        keep = (idx != padding_idx).unsqueeze(-1)
        y = torch.where(keep, y, y.detach())
    return y


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


def bpr_triplet(Ut, It, users, pos, neg):
    """mean(softplus(<u, i-> - <u, i+>)) over the triplets, fused (gathers, dots, criterion in one kernel each way):
    MF.fit (MF-BPR/main.py:81-93) for one negative per positive.  = torch.ops.recengine.bpr_triplet(...)[0]."""
    return _R.bpr_triplet(Ut, It, users, pos, neg)[0]


class BPRTripletLoss(torch.nn.Module):
    def forward(self, Ut, It, users, pos, neg):
        return bpr_triplet(Ut, It, users, pos, neg)


def score_full(Q, E):
    """scores[b, n] = <Q[b], E[n]> as exact k-ordered fp32 chains (`recommend_from_full`); differentiable (CE over the catalog).
    = torch.ops.recengine.score_dense."""
    return _R.score_dense(Q, E)


def spmm_sym(crow, col, val, X, plan=None):
    """A @ X for a SYMMETRIC CSR matrix (LightGCN's normalised bipartite adjacency): the backward is the same kernel.
    = torch.ops.recengine.spmm_csr (the row-order plan is built once per adjacency and cached)."""
    return _R.spmm_csr(crow, col, val, X)


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


def linear(x, weight, bias=None):
    """x W^T (+ b) on the engine's fp32 MFMA GEMM (re_gemm_f32), differentiable: torch.nn.functional.linear for 2-D x."""
    y = _R.gemm(x.contiguous(), weight.contiguous(), False, True)
    return y if bias is None else y + bias


class Linear(torch.nn.Module):
    """Drop-in for torch.nn.Linear(in_features, out_features, bias) on 2-D inputs; forward and both backward products are
    recengine::gemm launches (nn.Linear in the reference's MLP blocks: DeepFM/main.py:103-124, DCN/main.py:48-69)."""

    def __init__(self, in_features, out_features, bias=True, device=None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = torch.nn.Parameter(torch.empty(out_features, in_features, device=device))
        self.bias = torch.nn.Parameter(torch.zeros(out_features, device=device)) if bias else None
        torch.nn.init.xavier_normal_(self.weight)

    def forward(self, x):
        return linear(x, self.weight, self.bias)


class _SpmmGeneral(torch.autograd.Function):
    """Y = A X for a CSR matrix A that is NOT symmetric: the backward is the same kernel on the CSR form of A^T (built once by the
    caller, e.g. with `csr_transpose`).  NGCF's left-normalised adjacency D^-1 (A + I) (NGCF/main.py:76-87)."""

    @staticmethod
    def forward(ctx, X, crow, col, val, t_crow, t_col, t_val):
        ctx.save_for_backward(t_crow, t_col, t_val)
        return _R.spmm_csr(crow, col, val, X.detach())

    @staticmethod
    def backward(ctx, dY):
        t_crow, t_col, t_val = ctx.saved_tensors
        return _R.spmm_csr(t_crow, t_col, t_val, dY.contiguous()), None, None, None, None, None, None


def spmm(A, At, X):
    """A @ X with A = (crow, col, val) and At = the CSR form of its transpose (`csr_transpose(A, n_cols)`)."""
    return _SpmmGeneral.apply(X, *A, *At)


def csr_transpose(A, n_cols):
    """(crow, col, val) of A^T, rows sorted, columns ascending inside a row.  Host-side, once per adjacency (like the reference's
    `to_adjacency`, NGCF/main.py:82-86)."""
    crow, col, val = A
    n_rows = crow.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n_rows, device=crow.device), crow[1:] - crow[:-1])
    order = torch.argsort(col * n_rows + rows)           # by (column, row)
    t_crow = torch.zeros(n_cols + 1, dtype=torch.int64, device=crow.device)
    t_crow[1:] = torch.cumsum(torch.bincount(col, minlength=n_cols), 0)
    return t_crow, rows[order].contiguous(), val[order].contiguous()


class GraphedStep:
    """One whole training step -- zero the gradients, `fit`, backward, optimizer step -- captured ONCE as a hipGraph and replayed per
    batch: a model file on the torch-op surface is a chain of small launches (a DCN step is ~150 of them) and the CPU launch path,
    not the GPU, sets the eager step time.  The engine's ops are stream-ordered and allocate through torch only, so they capture
    like aten's; what cannot be captured is a host synchronisation inside `fit` (boolean-mask indexing, `.item()`).

        step = GraphedStep(model, lambda x, y: sum(model.fit(x, y).values()), torch.optim.Adam(model.parameters(), capturable=True), (x0, y0))
        for x, y in batches: loss = step(x, y)      # `loss` is the graph's static output tensor: read it before the next call

    Batches must have the example's shapes and dtypes (a short last batch takes its own GraphedStep)."""

    def __init__(self, model, loss_fn, optimizer, example_inputs, warmup=3):
        self.static_in = tuple(t.clone() for t in example_inputs)
        self.optimizer = optimizer
        params = [p for g in optimizer.param_groups for p in g["params"]]
        # Everything the warm-up and capture steps change is snapshotted and put back afterwards: the parameters, the optimizer's
        # state as it stands (a Coach builds one GraphedStep per input shape on a SHARED optimizer -- a short last batch, or the
        # first step after load_checkpoint, arrives with live Adam moments and step counts), and the module's buffers
        # (BatchNorm running statistics / num_batches_tracked advance in the warm-up).
        keep = [p.detach().clone() for p in params]
        had_state = {p: (p in optimizer.state and len(optimizer.state[p]) > 0) for p in params}
        keep_state = {p: {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in optimizer.state[p].items()}
                      for p in params if had_state[p]}
        buffers = [b for b in model.buffers()] if isinstance(model, torch.nn.Module) else []
        keep_buf = [b.detach().clone() for b in buffers]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up on a side stream (allocator pools, lazy optimizer state)
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=True)
                loss_fn(*self.static_in).backward()
                optimizer.step()
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with recording(self.graph):
            self.loss = loss_fn(*self.static_in)
            self.loss.backward()
            optimizer.step()
        with torch.no_grad():                          # roll the warm-up and capture steps back, IN PLACE (the graph holds these tensors)
            for p, k in zip(params, keep):
                p.copy_(k)
            for b, k in zip(buffers, keep_buf):
                # only what the warm-up CHANGED (BatchNorm statistics, counters): an in-place write bumps a tensor's version, and the ops' plan
                # caches are keyed on it (torch_ops._plan: a rewritten adjacency gets a new plan and the old one is freed) -- writing an
                # unchanged `crow` back here made the next evaluation rebuild the SpMM plan and free the one whose address THIS graph replays
                # (a memory access fault two epochs later, or silently another tensor's bytes)
                if not torch.equal(b, k):
                    b.copy_(k)
            for p in params:
                st = optimizer.state.get(p, {})
                for name, v in st.items():
                    if not torch.is_tensor(v):
                        continue
                    if had_state[p] and name in keep_state[p] and torch.is_tensor(keep_state[p][name]):
                        v.copy_(keep_state[p][name])
                    else:
                        v.zero_()                      # state the warm-up created: as a fresh optimizer's

    def __call__(self, *inputs):
        for s, t in zip(self.static_in, inputs):
            s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.loss
