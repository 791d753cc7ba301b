"""Oracle: SASRec encode / fit / recommend_from_full (TEST INFRASTRUCTURE).  torch-CPU fp32.

Follows SASRec/main.py of the reference, decomposed into explicit matrix math (no
nn.MultiheadAttention / nn.Conv1d modules) so that every intermediate the fused HIP
kernels produce has a named counterpart here:

  encode                SASRec/main.py:178-193  (embedding*sqrt(D) + positions, dropout, pad-mask, blocks, lastLN)
  mark_position         SASRec/main.py:159-161  (absolute positions 0..S-1, regardless of padding)
  after_one_block       SASRec/main.py:163-176  Q = LN_a(x); x = MHA(Q, x, x, causal) + x; y = LN_f(x);
                                                x = FFN(y) [residual onto the layer-normed y]; x[pad] = 0
  PointWiseFeedForward  SASRec/main.py:31-50    conv2(relu(dropout1(conv1(y)))) -> dropout2 -> + y
  fit                   SASRec/main.py:195-221  BCE / BPR / CE over the M non-pad positions
  recommend_from_full   SASRec/main.py:223-228  scores = u[:, -1, :] @ E[1:]^T

Quirks kept on purpose: K,V are NOT layer-normed; left-pad positions ARE attended as keys;
LN eps = 1e-8; IPos/INeg are 0-based into E[1:].

Parameters are passed as a dict keyed by the reference's state_dict names
(`Item.embeddings.weight`, `attnLayers.0.in_proj_weight`, `fwdLayers.0.conv1.weight` [D,D,1], ...).

Dropout: `drop=None` (p = 0, what the golden vectors use) or `drop=dict(p=, seed=)` which applies the
ENGINE's counter-based masks (oracle/rng.py) at the reference's five dropout sites.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import criterions, rng


def _mask(drop, stream, shape):
    if not drop or drop.get("p", 0.0) <= 0.0:
        return None
    keep = rng.keep_mask(drop["seed"], stream, shape, drop["p"])
    return torch.from_numpy(keep.astype(np.float32)) / (1.0 - drop["p"])


def _apply(x, m):
    return x if m is None else x * m


def layer_norm(x, w, b, eps=1e-8):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def block(x, pad, P, l, drop=None, gates=None):
    """One SASRec block on x [B,S,D]; pad [B,S,1] bool.  SASRec/main.py:163-176.
    gates (optional): {l: (open [B,S,D] bool, eps)} -- where the FFN's pre-activation is within eps of zero (a relu kink) the gate
    is taken from `open` (the implementation under test) instead of the sign computed here: there the two sides' rounding decides,
    and a flipped gate changes that element's gradient by its whole upstream value, which no tolerance covers.  A third tuple element (a
    dict) receives {window: real-row entries inside the window, total: real-row entries, mismatch_outside: real-row entries OUTSIDE the
    window where `open` differs from this side's sign} -- the caller asserts a ceiling on the first and zero for the last."""
    B, S, D = x.shape
    pre = f"attnLayers.{l}."
    Wi, bi = P[pre + "in_proj_weight"], P[pre + "in_proj_bias"]
    Wq, Wk, Wv = Wi[0:D], Wi[D:2 * D], Wi[2 * D:3 * D]
    bq, bk, bv = bi[0:D], bi[D:2 * D], bi[2 * D:3 * D]
    q_in = layer_norm(x, P[f"attnLNs.{l}.weight"], P[f"attnLNs.{l}.bias"])
    q = q_in @ Wq.T + bq
    k = x @ Wk.T + bk
    v = x @ Wv.T + bv
    scores = (q @ k.transpose(1, 2)) / math.sqrt(D)          # single head: head_dim = D
    causal = torch.ones(S, S, dtype=torch.bool).triu(1)
    scores = scores.masked_fill(causal, float("-inf"))
    A = torch.softmax(scores, dim=-1)
    A = _apply(A, _mask(drop, rng.stream_attn(l), (B, S, S)))
    attn = (A @ v) @ P[pre + "out_proj.weight"].T + P[pre + "out_proj.bias"]
    x = attn + x
    y = layer_norm(x, P[f"fwdLNs.{l}.weight"], P[f"fwdLNs.{l}.bias"])
    W1, b1 = P[f"fwdLayers.{l}.conv1.weight"].squeeze(-1), P[f"fwdLayers.{l}.conv1.bias"]
    W2, b2 = P[f"fwdLayers.{l}.conv2.weight"].squeeze(-1), P[f"fwdLayers.{l}.conv2.bias"]
    h = y @ W1.T + b1
    h = _apply(h, _mask(drop, rng.stream_ffn1(l), (B, S, D)))
    if gates is not None and l in gates:
        open_, eps = gates[l][0], gates[l][1]
        near = h.detach().abs() < eps
        if len(gates[l]) > 2:      # a report for the caller to assert on: how many gates were borrowed, and whether the others agree
            live = ~pad.expand_as(near)
            nz = h.detach() != 0           # (an element dropout has zeroed is exactly 0 on both sides: its gate multiplies 0 forward and backward)
            gates[l][2].update(window=int((near & live & nz).sum()), total=int((live & nz).sum()),
                               mismatch_outside=int(((open_ != (h.detach() > 0)) & ~near & live).sum()))
        h = h * torch.where(near, open_, h.detach() > 0).to(h.dtype)
    else:
        h = torch.relu(h)
    o = h @ W2.T + b2
    o = _apply(o, _mask(drop, rng.stream_ffn2(l), (B, S, D)))
    x = o + y
    return x.masked_fill(pad, 0.0)


def encode(P, seq, num_blocks=2, drop=None, gates=None):
    """-> (userEmbds [B,S,D], itemEmbds = E[1:] [N,D]).  SASRec/main.py:178-193."""
    E = P["Item.embeddings.weight"]
    B, S = seq.shape
    D = E.shape[1]
    pad = (seq == 0).unsqueeze(-1)
    x = E[seq] * (D ** 0.5)
    x = x + P["Position.weight"][:S].unsqueeze(0)
    x = _apply(x, _mask(drop, rng.STREAM_EMBED, (B, S, D)))
    x = x.masked_fill(pad, 0.0)
    for l in range(num_blocks):
        x = block(x, pad, P, l, drop, gates)
    u = layer_norm(x, P["lastLN.weight"], P["lastLN.bias"])
    return u, E[1:]


def fit(P, seq, pos, neg, loss="BCE", num_blocks=2, drop=None, gates=None):
    """-> rec_loss scalar.  SASRec/main.py:195-221."""
    u, items = encode(P, seq, num_blocks, drop, gates)
    idx = seq != 0
    u = u[idx]
    if loss in ("BCE", "BPR"):
        pl = (u * items[pos[idx]]).sum(-1)
        nl = (u * items[neg[idx]]).sum(-1)
        if loss == "BCE":
            return criterions.bce_with_logits(pl, torch.ones_like(pl)) + \
                criterions.bce_with_logits(nl, torch.zeros_like(nl))
        return criterions.bpr_loss(pl, nl)
    logits = u @ items.T
    return criterions.cross_entropy(logits, pos[idx])


def recommend_from_full(P, seq, num_blocks=2):
    u, items = encode(P, seq, num_blocks)
    return u[:, -1, :] @ items.T


def params_from_npz(z, requires_grad=False):
    P = {}
    for k in z.files:
        if k.startswith("param/") and z[k].dtype == np.float32:
            t = torch.from_numpy(z[k].copy())
            t.requires_grad_(requires_grad)
            P[k[len("param/"):]] = t
    return P
