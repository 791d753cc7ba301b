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


def _scalars(step, lr, b1, b2, hyper):
    """-> (step size, 1 / sqrt(bias correction 2)) as float32, or None for "leave everything as it is".  hyper = the two scalars themselves
    (the engine's captured steps read them from device memory: re_adam_step_dev); {0, 0} there gates the whole update off."""
    f = np.float32
    if hyper is not None:
        ss, ib = f(hyper[0]), f(hyper[1])
        return None if ib == 0 else (ss, ib)
    return f(lr / (1.0 - b1 ** step)), f(1.0 / math.sqrt(1.0 - b2 ** step))


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0, hyper=None):
    """In-place on float32 numpy arrays; `step` is the 1-based step count."""
    f = np.float32
    if hyper is not None:
        sc = _scalars(step, lr, b1, b2, hyper)
        if sc is None:
            return p, m, v
        g = g + f(wd) * p if wd != 0.0 else g
        m[...] = f(b1) * m + f(1.0 - b1) * g
        v[...] = f(b2) * v + f(1.0 - b2) * g * g
        p[...] = p - sc[0] * (m / (np.sqrt(v) * sc[1] + f(eps)))
        return p, m, v
    g = g + f(wd) * p if wd != 0.0 else g
    m[...] = f(b1) * m + f(1.0 - b1) * g
    v[...] = f(b2) * v + f(1.0 - b2) * g * g
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    step_size = f(lr / bc1)
    denom = np.sqrt(v) / f(math.sqrt(bc2)) + f(eps)
    p[...] = p - step_size * (m / denom)
    return p, m, v


def sparse_adam_rows(W, m, v, idx, g, step, lr, b1=0.9, b2=0.999, eps=1e-8, wd=0.0, padding_idx=-1, hyper=None):
    """Row-sparse Adam on float32 numpy arrays, in place: the rows of (W, m, v) that idx points at are updated with the sum of
    their gradient rows (added in position order); all other rows are left alone.  `torch.optim.SparseAdam`'s rule
    (torch/optim/sparse_adam.py: moments updated at the gradient's indices only, bias corrections from the global step) with
    the dense optimizer's coupled weight decay applied on the touched rows.  Pinned against torch.optim.SparseAdam (wd = 0)
    in tests/test_oracle_golden.py."""
    f = np.float32
    if hyper is not None:
        sc = _scalars(step, lr, b1, b2, hyper)
        if sc is None:
            return W, m, v
        step, lr = 1, float(sc[0]) * (1.0 - b1)                      # (re-expressed through the formulas below: step size and ...
        b2_eff = 1.0 - 1.0 / float(sc[1]) ** 2                       #  ... bias correction 2 as given)
    R = W.shape[0]
    idx = np.asarray(idx).reshape(-1)
    g = np.asarray(g, dtype=np.float32).reshape(idx.size, -1)
    G = {}
    for i, r in enumerate(idx.tolist()):
        if r == padding_idx or r < 0 or r >= R:
            continue
        G[r] = g[i].copy() if r not in G else (G[r] + g[i]).astype(np.float32)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step if hyper is None else 1.0 - b2_eff
    step_size = f(lr / bc1)
    for r, gr in G.items():
        gg = gr + f(wd) * W[r] if wd != 0.0 else gr
        m[r] = f(b1) * m[r] + f(1.0 - b1) * gg
        v[r] = f(b2) * v[r] + f(1.0 - b2) * gg * gg
        W[r] = W[r] - step_size * (m[r] / (np.sqrt(v[r]) / f(math.sqrt(bc2)) + f(eps)))
    return W, m, v
