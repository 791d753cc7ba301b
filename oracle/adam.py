"""Oracle: dense Adam with coupled L2 weight decay (TEST INFRASTRUCTURE).

`torch.optim.Adam(params, lr, betas, weight_decay)` as configured by the reference's cfg dump
(benchmark/Amazon2014Beauty_550_LOU/SASRec.json:254-300: optimizer adam, lr 5e-4, betas (0.9, 0.999), weight_decay 1e-6).
Dense semantics: every row moves every step (moment decay + L2), also rows whose gradient is zero.
Single-tensor torch formulation (torch/optim/adam.py `_single_tensor_adam`, eps = 1e-8, no amsgrad):
    g = g + wd * p;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g
    p = p - (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
"""
import math

import numpy as np


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """In-place on float32 numpy arrays; `step` is the 1-based step count."""
    f = np.float32
    g = g + f(wd) * p if wd != 0.0 else g
    m[...] = f(b1) * m + f(1.0 - b1) * g
    v[...] = f(b2) * v + f(1.0 - b2) * g * g
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = f(lr / bc1)
    denom = np.sqrt(v) / f(math.sqrt(bc2)) + f(eps)
    p[...] = p - step_size * (m / denom)
    return p, m, v


def sparse_adam_rows(W, m, v, idx, g, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0, padding_idx=-1):
    """Row-sparse Adam on float32 numpy arrays, in place: the rows of (W, m, v) that idx points at are updated with the sum of
    their gradient rows (added in position order); all other rows are left alone.  `torch.optim.SparseAdam`'s rule
    (torch/optim/sparse_adam.py: moments updated at the gradient's indices only, bias corrections from the global step) with
    the dense optimizer's coupled weight decay applied on the touched rows.  Pinned against torch.optim.SparseAdam (wd = 0)
    in tests/test_oracle_golden.py."""
    f = np.float32
    R = W.shape[0]
    idx = np.asarray(idx).reshape(-1)
    g = np.asarray(g, dtype=np.float32).reshape(idx.size, -1)
    G = {}
    for i, r in enumerate(idx.tolist()):
        if r == padding_idx or r < 0 or r >= R:
            continue
        G[r] = g[i].copy() if r not in G else (G[r] + g[i]).astype(np.float32)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = f(lr / bc1)
    for r, gr in G.items():
        gg = gr + f(wd) * W[r] if wd != 0.0 else gr
        m[r] = f(b1) * m[r] + f(1.0 - b1) * gg
        v[r] = f(b2) * v[r] + f(1.0 - b2) * gg * gg
        W[r] = W[r] - step_size * (m[r] / (np.sqrt(v[r]) / f(math.sqrt(bc2)) + f(eps)))
    return W, m, v
