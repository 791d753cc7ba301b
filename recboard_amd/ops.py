"""Torch-tensor front end of the C ABI (include/recengine.h): argument checks, buffer allocation, stream plumbing.

PyTorch is used for device memory and streams only; every function here ends in exactly one (or a fixed few)
`librecengine.so` calls on `torch.cuda.current_stream()`.  CPU tensors are rejected -- there is no CPU fallback.
"""
import ctypes

import torch

from . import lib

LOSS_BCE, LOSS_BPR = 0, 1
TOPK_MAX = 64


# While one of this package's captures records (recboard_amd.capture.recording), every tensor whose ADDRESS goes into a launch is noted here:
# a hipGraph replays raw addresses, so the graph has to own what it was handed (the storage, not the tensor: no autograd history is kept
# alive) -- a workspace, plan or view whose last Python reference goes away after the capture would otherwise be recycled by torch's
# allocator (silently another tensor's bytes) or, once any later capture's `empty_cache()` has returned its block to the driver, be
# unmapped under the replay ("Memory access fault by GPU", round 5: DESIGN.md section 8.0).
_KEEP = None


def _note(t):
    if _KEEP is not None:
        _KEEP.append(t.untyped_storage())
    return t.data_ptr()


def _p(t):
    return None if t is None else ctypes.c_void_p(_note(t))


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t, dtype, name, contiguous=True):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"recengine: `{name}` must be a tensor on a HIP device (no CPU fallback exists)")
    if t.dtype != dtype:
        raise TypeError(f"recengine: `{name}` must be {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise ValueError(f"recengine: `{name}` must be contiguous")
    return t


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------------ K1
def gather_rows(W: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[..., :] = W[idx[...], :]   (re_gather_rows)."""
    _req(W, torch.float32, "W")
    _req(idx, torch.int64, "idx")
    R, D = W.shape
    out = torch.empty(idx.shape + (D,), dtype=torch.float32, device=W.device)
    lib.check(lib.load().re_gather_rows(_p(W), R, D, _p(idx), idx.numel(), _p(out), _stream()), "re_gather_rows")
    return out


def sasrec_embed(E, P, seq, scale, drop_p=0.0, seed=0, out=None, seed_dev=None):
    """(E[seq]*scale + P[s]) with pad rows zeroed and optional engine dropout (re_sasrec_embed)."""
    _req(E, torch.float32, "E"); _req(P, torch.float32, "P"); _req(seq, torch.int64, "seq")
    B, S = seq.shape
    R, D = E.shape
    if P.shape[0] < S or P.shape[1] != D:
        raise ValueError("recengine: position table must be [>=S, D]")
    if out is None:
        out = torch.empty((B, S, D), dtype=torch.float32, device=E.device)
    lib.check(lib.load().re_sasrec_embed(_p(E), R, D, _p(P), _p(seq), B, S, float(scale), float(drop_p),
                                         int(seed) & 0xFFFFFFFF, _p(seed_dev), _p(out), _stream()), "re_sasrec_embed")
    return out


def sasrec_embed_bwd(gx, seq, scale, drop_p, seed, dP, ws=None, seed_dev=None):
    """In place on gx [B,S,D] (gradient w.r.t. x0 -> scatter contribution rows); writes dP [S,D] (re_sasrec_embed_bwd)."""
    _req(gx, torch.float32, "gx"); _req(seq, torch.int64, "seq"); _req(dP, torch.float32, "dP")
    B, S, D = gx.shape
    L = lib.load()
    if ws is None:
        ws = _ws(L.re_sasrec_embed_bwd_workspace_bytes(S, D), gx.device)
    lib.check(L.re_sasrec_embed_bwd(_p(gx), _p(seq), B, S, D, float(scale), float(drop_p), int(seed) & 0xFFFFFFFF, _p(seed_dev), _p(dP),
                                    _p(ws), ws.numel(), _stream()), "re_sasrec_embed_bwd")
    return gx


def scatter_add_rows(g: torch.Tensor, idx: torch.Tensor, R: int, padding_idx: int = -1, scale: float = 1.0, out=None,
                     ws=None, accumulate=False):
    """Dense [R, D] gradient of gather_rows, deterministic (re_scatter_add_rows)."""
    _req(g, torch.float32, "g"); _req(idx, torch.int64, "idx")
    D = g.shape[-1]
    n = idx.numel()
    if g.numel() != n * D:
        raise ValueError("recengine: g must have one row per index")
    L = lib.load()
    if ws is None:
        ws = _ws(L.re_scatter_add_rows_workspace_bytes(n, D, R), g.device)
    dW = out if out is not None else torch.empty((R, D), dtype=torch.float32, device=g.device)
    _req(dW, torch.float32, "out")
    if accumulate and out is None:
        raise ValueError("recengine: accumulate=True needs `out`")
    lib.check(L.re_scatter_add_rows(_p(g), _p(idx), n, D, R, int(padding_idx), float(scale), _p(dW), int(bool(accumulate)),
                                    _p(ws), ws.numel(), _stream()), "re_scatter_add_rows")
    return dW


def sparse_adam_rows(g, idx, W, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, padding_idx=-1, ws=None):
    """Row-sparse Adam (re_sparse_adam_rows): rows of (W, m, v) hit by idx are updated in place with the summed gradient rows."""
    _req(g, torch.float32, "g"); _req(idx, torch.int64, "idx")
    for t, nme in ((W, "W"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nme)
    R, D = W.shape
    n = idx.numel()
    if g.numel() != n * D:
        raise ValueError("recengine: g must have one row per index")
    L = lib.load()
    if ws is None:
        ws = _ws(L.re_scatter_add_rows_workspace_bytes(n, D, R), g.device)
    lib.check(L.re_sparse_adam_rows(_p(g), _p(idx), n, D, R, int(padding_idx), _p(W), _p(m), _p(v), int(step), float(lr), float(beta1),
                                    float(beta2), float(eps), float(weight_decay), _p(ws), ws.numel(), _stream()), "re_sparse_adam_rows")


def sparse_adam_rows_dev(g, idx, W, m, v, hyper, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, padding_idx=-1, ws=None):
    """sparse_adam_rows with the step-dependent scalars in device memory (re_sparse_adam_rows_dev; for captured steps)."""
    _req(g, torch.float32, "g"); _req(idx, torch.int64, "idx"); _req(hyper, torch.float32, "hyper")
    for t, nme in ((W, "W"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nme)
    R, D = W.shape
    n = idx.numel()
    L = lib.load()
    if ws is None:
        ws = _ws(L.re_scatter_add_rows_workspace_bytes(n, D, R), g.device)
    lib.check(L.re_sparse_adam_rows_dev(_p(g), _p(idx), n, D, R, int(padding_idx), _p(W), _p(m), _p(v), _p(hyper), float(beta1), float(beta2),
                                        float(eps), float(weight_decay), _p(ws), ws.numel(), _stream()), "re_sparse_adam_rows_dev")


def sparse_adam_small_ok(keys, W):
    """Whether re_sparse_adam_rows_small takes this update: D = 64 / 128 and a key list every workgroup can afford to scan."""
    return W.shape[1] in (64, 128) and keys.numel() <= (1 << 15) and W.shape[0] < 0xFFFFFFFE


def sparse_adam_rows_small(g, keys, W, m, v, step=0, lr=0.0, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, padding_idx=-1, hyper=None,
                          n_dev=None, n_mul=1):
    """Row-sparse Adam for a small key list in one launch (re_sparse_adam_rows_small).  keys: int32 or int64, [n] or [regions, stride]
    (one row of g per entry); n_dev (device int32[1]): only the first n_dev * n_mul entries of every region are read; hyper (device
    float32[2]) replaces the host's step / lr (captured steps)."""
    _req(g, torch.float32, "g")
    if keys.dtype not in (torch.int32, torch.int64) or not keys.is_cuda or not keys.is_contiguous():
        raise TypeError("recengine: `keys` must be a contiguous int32 or int64 tensor on the HIP device")
    for t, nme in ((W, "W"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nme)
    R, D = W.shape
    regions, stride = (1, keys.numel()) if keys.dim() == 1 else (keys.shape[0], keys.shape[1])
    if g.numel() != keys.numel() * D:
        raise ValueError("recengine: g must have one row per key entry")
    if hyper is not None:
        _req(hyper, torch.float32, "hyper")
    if n_dev is not None:
        _req(n_dev, torch.int32, "n_dev")
    lib.check(lib.load().re_sparse_adam_rows_small(_p(g), _p(keys), keys.element_size(), int(regions), int(stride), _p(n_dev), int(n_mul),
                                                   int(stride), D, R, int(padding_idx), _p(W), _p(m), _p(v), _p(hyper), int(step), float(lr),
                                                   float(beta1), float(beta2), float(eps), float(weight_decay), _stream()),
              "re_sparse_adam_rows_small")


def scatter_workspace(n, D, R, device):
    """The workspace of scatter_plan / scatter_apply for n indices and rows of D floats (a plan made with D serves applies of any D' <= D)."""
    return _ws(lib.load().re_scatter_add_rows_workspace_bytes(int(n), int(D), int(R)), device)


def scatter_plan(idx, D, R, ws, padding_idx=-1, zero=None):
    """Index half of scatter_add_rows (re_scatter_plan): sorts (destination row, position) into `ws`; optionally zero-fills
    `zero` (the table scatter_apply will accumulate into).  Depends on idx only -- may run on a side stream."""
    _req(idx, torch.int64, "idx"); _req(ws, torch.uint8, "ws")
    if zero is not None:
        _req(zero, torch.float32, "zero")
    lib.check(lib.load().re_scatter_plan(_p(idx), idx.numel(), int(D), int(R), int(padding_idx), _p(zero),
                                         0 if zero is None else zero.numel(), _p(ws), ws.numel(), _stream()), "re_scatter_plan")


def scatter_apply(g, R, out, ws, scale=1.0, accumulate=True):
    """Data half of scatter_add_rows (re_scatter_apply): out (+)= segmented sum of g's rows in the order scatter_plan left in ws.
    accumulate="rows": only the rows that occur are written (assigned); the others keep their contents (no zero fill)."""
    _req(g, torch.float32, "g"); _req(out, torch.float32, "out"); _req(ws, torch.uint8, "ws")
    n, D = g.shape
    lib.check(lib.load().re_scatter_apply(_p(g), n, D, int(R), float(scale), _p(out), 2 if accumulate == "rows" else int(bool(accumulate)), _p(ws), ws.numel(),
                                          _stream()), "re_scatter_apply")
    return out


# ------------------------------------------------------------------------------------------------ K3
def pair_loss_fwd(U, E, pos, neg, valid, kind, e_off=0):
    """U [n, D] (rows may be strided), E [R, D]; returns (loss[1], logits[n,2], count int32[1])."""
    _req(U, torch.float32, "U", contiguous=False); _req(E, torch.float32, "E")
    _req(pos, torch.int64, "pos"); _req(neg, torch.int64, "neg")
    if U.dim() != 2 or U.stride(1) != 1:
        raise ValueError("recengine: U must be [n, D] with unit inner stride")
    n, D = U.shape
    if valid is not None:
        _req(valid, torch.uint8, "valid")
    L = lib.load()
    dev = U.device
    logits = torch.empty((n, 2), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    count = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = _ws(L.re_pair_loss_workspace_bytes(n), dev)
    lib.check(L.re_pair_loss_fwd(_p(U), U.stride(0), _p(E), E.shape[0], D, e_off, _p(pos), _p(neg), _p(valid), n,
                                 kind, _p(logits), _p(loss), _p(count), _p(ws), ws.numel(), _stream()),
              "re_pair_loss_fwd")
    return loss, logits, count


def pair_loss_bwd(U, E, pos, neg, valid, kind, logits, count, dloss, e_off=0, out=None):
    """-> (dU [n,D], gpos [n,D], gneg [n,D]) contribution rows."""
    n, D = U.shape
    dev = U.device
    if out is not None:
        dU, gpos, gneg = (_req(t, torch.float32, "out") for t in out)
    else:
        dU = torch.empty((n, D), dtype=torch.float32, device=dev)
        gpos = torch.empty((n, D), dtype=torch.float32, device=dev)
        gneg = torch.empty((n, D), dtype=torch.float32, device=dev)
    if dloss is not None:
        _req(dloss, torch.float32, "dloss")
    lib.check(lib.load().re_pair_loss_bwd(_p(U), U.stride(0), _p(E), E.shape[0], D, e_off, _p(pos), _p(neg), _p(valid),
                                          n, kind, _p(logits), _p(count), _p(dloss), _p(dU), D, _p(gpos), _p(gneg),
                                          _stream()), "re_pair_loss_bwd")
    return dU, gpos, gneg


def pair_loss_fwd_bwd(U, E, pos, neg, valid, kind, count, e_off=0, out=None):
    """Training form of the pair criteria: loss and its gradient rows in one pass (re_pair_loss_fwd_bwd).  `count` is the
    DEVICE int32[1] number of valid positions (batch assembly).  -> (loss[1], dU, gpos, gneg)."""
    _req(U, torch.float32, "U", contiguous=False); _req(E, torch.float32, "E")
    _req(pos, torch.int64, "pos"); _req(neg, torch.int64, "neg"); _req(count, torch.int32, "count")
    if U.dim() != 2 or U.stride(1) != 1:
        raise ValueError("recengine: U must be [n, D] with unit inner stride")
    n, D = U.shape
    if valid is not None:
        _req(valid, torch.uint8, "valid")
    dev = U.device
    if out is not None:
        dU, gpos, gneg = (_req(t, torch.float32, "out") for t in out)
    else:
        dU, gpos, gneg = (torch.empty((n, D), dtype=torch.float32, device=dev) for _ in range(3))
    L = lib.load()
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    ws = _ws(L.re_pair_loss_workspace_bytes(n), dev)
    lib.check(L.re_pair_loss_fwd_bwd(_p(U), U.stride(0), _p(E), E.shape[0], D, e_off, _p(pos), _p(neg), _p(valid), n, kind, _p(count),
                                     _p(loss), _p(dU), D, _p(gpos), _p(gneg), _p(ws), ws.numel(), _stream()), "re_pair_loss_fwd_bwd")
    return loss, dU, gpos, gneg


def bpr_triplet_fwd_bwd(Ut, It, users, pos, neg):
    """MF-BPR / LightGCN training form: mean BPR loss and the three gradient-row sets in one pass.  -> (loss[1], gu, gp, gn)."""
    _req(Ut, torch.float32, "Ut"); _req(It, torch.float32, "It")
    for t, nme in ((users, "users"), (pos, "pos"), (neg, "neg")):
        _req(t, torch.int64, nme)
    n = users.numel()
    if pos.numel() != n or neg.numel() != n:
        raise ValueError("recengine: one positive and one negative per user row (K = 1)")
    D = Ut.shape[1]
    L = lib.load()
    dev = Ut.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    gu, gp, gn = (torch.empty((n, D), dtype=torch.float32, device=dev) for _ in range(3))
    ws = _ws(L.re_pair_loss_workspace_bytes(n), dev)
    lib.check(L.re_bpr_triplet_fwd_bwd(_p(Ut), Ut.shape[0], _p(It), It.shape[0], D, _p(users), _p(pos), _p(neg), n, _p(loss), _p(gu),
                                       _p(gp), _p(gn), _p(ws), ws.numel(), _stream()), "re_bpr_triplet_fwd_bwd")
    return loss, gu, gp, gn


def bpr_triplet_step_rows(Ut, It, users, pos, neg, g=None, keys=None):
    """bpr_triplet_fwd_bwd with the gradient rows as ONE [3, n, D] block and their int32 destination rows in the user | item arena
    ([3, n]; item rows offset by the number of users; -1 = none): the inputs of scatter_add_rows_small(..., n_regions=3, adam=...).
    -> (loss[1], g, keys)."""
    _req(Ut, torch.float32, "Ut"); _req(It, torch.float32, "It")
    for t, nme in ((users, "users"), (pos, "pos"), (neg, "neg")):
        _req(t, torch.int64, nme)
    n = users.numel()
    if pos.numel() != n or neg.numel() != n:
        raise ValueError("recengine: one positive and one negative per user row (K = 1)")
    D = Ut.shape[1]
    L = lib.load()
    dev = Ut.device
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    if g is None:
        g = torch.empty((3, n, D), dtype=torch.float32, device=dev)
    if keys is None:
        keys = torch.empty((3, n), dtype=torch.int32, device=dev)
    _req(g, torch.float32, "g"); _req(keys, torch.int32, "keys")
    if g.numel() != 3 * n * D or keys.numel() != 3 * n:
        raise ValueError("recengine: bpr_triplet_step_rows buffer shapes")
    ws = _ws(L.re_pair_loss_workspace_bytes(n), dev)
    lib.check(L.re_bpr_triplet_step_rows(_p(Ut), Ut.shape[0], _p(It), It.shape[0], D, _p(users), _p(pos), _p(neg), n, _p(loss), _p(g), _p(keys),
                                         _p(ws), ws.numel(), _stream()), "re_bpr_triplet_step_rows")
    return loss, g, keys


def bpr_triplet_fwd(Ut, It, users, pos, neg):
    _req(Ut, torch.float32, "Ut"); _req(It, torch.float32, "It")
    for t, nme in ((users, "users"), (pos, "pos"), (neg, "neg")):
        _req(t, torch.int64, nme)
    n = users.numel()
    if pos.numel() != n or neg.numel() != n:
        raise ValueError("recengine: one positive and one negative per user row (K = 1)")
    D = Ut.shape[1]
    L = lib.load()
    dev = Ut.device
    logits = torch.empty((n, 2), dtype=torch.float32, device=dev)
    loss = torch.empty((1,), dtype=torch.float32, device=dev)
    ws = _ws(L.re_pair_loss_workspace_bytes(n), dev)
    lib.check(L.re_bpr_triplet_fwd(_p(Ut), Ut.shape[0], _p(It), It.shape[0], D, _p(users), _p(pos), _p(neg), n,
                                   _p(logits), _p(loss), _p(ws), ws.numel(), _stream()), "re_bpr_triplet_fwd")
    return loss, logits


def bpr_triplet_bwd(Ut, It, users, pos, neg, logits, dloss):
    n = users.numel()
    D = Ut.shape[1]
    dev = Ut.device
    gu, gp, gn = (torch.empty((n, D), dtype=torch.float32, device=dev) for _ in range(3))
    lib.check(lib.load().re_bpr_triplet_bwd(_p(Ut), Ut.shape[0], _p(It), It.shape[0], D, _p(users), _p(pos), _p(neg), n,
                                            _p(logits), _p(dloss), _p(gu), _p(gp), _p(gn), _stream()),
              "re_bpr_triplet_bwd")
    return gu, gp, gn


# ------------------------------------------------------------------------------------------------ K4
def score_dense(Q, E):
    _req(Q, torch.float32, "Q"); _req(E, torch.float32, "E")
    B, D = Q.shape
    N = E.shape[0]
    out = torch.empty((B, N), dtype=torch.float32, device=Q.device)
    lib.check(lib.load().re_score_dense(_p(Q), _p(E), B, N, D, _p(out), _stream()), "re_score_dense")
    return out


def score_prepare(E):
    """Split the item table once for `score_topk(..., prep=)` (many user batches against one table).  -> opaque uint8 tensor;
    only D = 64 / 128 have a split form (others: returns None and score_topk takes its exact path)."""
    _req(E, torch.float32, "E")
    N, D = E.shape
    if D not in (64, 128) or N == 0:
        return None
    L = lib.load()
    prep = torch.empty(L.re_score_prepare_bytes(N, D), dtype=torch.uint8, device=E.device)
    lib.check(L.re_score_prepare(_p(E), N, D, _p(prep), prep.numel(), _stream()), "re_score_prepare")
    return prep


def score_topk(Q, E, seen_ptr, seen_idx, K, prep=None):
    """Fused score + seen-mask + top-K.  seen_ptr int64[B+1], seen_idx int64[nnz] with every user's ids ASCENDING
    (None/None = retain_seen).  -> (vals f32 [B,K] descending, idx int64 [B,K]).  prep = score_prepare(E) (optional)."""
    _req(Q, torch.float32, "Q"); _req(E, torch.float32, "E")
    B, D = Q.shape
    N = E.shape[0]
    if seen_ptr is not None:
        _req(seen_ptr, torch.int64, "seen_ptr"); _req(seen_idx, torch.int64, "seen_idx")
        if seen_ptr.numel() != B + 1:
            raise ValueError("recengine: seen_ptr must have B+1 entries")
    if not (0 < K <= TOPK_MAX):
        raise ValueError(f"recengine: K must be in 1..{TOPK_MAX}")
    if seen_idx is not None and seen_idx.numel() == 0:
        seen_ptr = seen_idx = None          # (nobody has seen anything: an empty tensor has no address to hand over)
    L = lib.load()
    dev = Q.device
    vals = torch.empty((B, K), dtype=torch.float32, device=dev)
    idx = torch.empty((B, K), dtype=torch.int64, device=dev)
    if prep is not None:
        ws = _ws(L.re_score_topk_prepared_workspace_bytes(B, N, D, K), dev)
        lib.check(L.re_score_topk_prepared(_p(Q), _p(E), _p(prep), B, N, D, _p(seen_ptr), _p(seen_idx), K, _p(vals), _p(idx),
                                           _p(ws), ws.numel(), _stream()), "re_score_topk_prepared")
        return vals, idx
    ws = _ws(L.re_score_topk_workspace_bytes(B, N, D, K), dev)
    lib.check(L.re_score_topk(_p(Q), _p(E), B, N, D, _p(seen_ptr), _p(seen_idx), K, _p(vals), _p(idx), _p(ws),
                              ws.numel(), _stream()), "re_score_topk")
    return vals, idx


# ------------------------------------------------------------------------------------------------ K6/K7
BLOCK_PARAM_ORDER = ("attnLNs.{l}.weight", "attnLNs.{l}.bias", "attnLayers.{l}.in_proj_weight", "attnLayers.{l}.in_proj_bias",
                     "attnLayers.{l}.out_proj.weight", "attnLayers.{l}.out_proj.bias", "fwdLNs.{l}.weight", "fwdLNs.{l}.bias",
                     "fwdLayers.{l}.conv1.weight", "fwdLayers.{l}.conv1.bias", "fwdLayers.{l}.conv2.weight",
                     "fwdLayers.{l}.conv2.bias")


def _ptr_table(tensors):
    """HOST array of device pointers (the ABI's `const float* const*`)."""
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        _req(t, torch.float32, f"param[{i}]")
        arr[i] = _note(t)
    return arr


def sasrec_block_tensors(named, L):
    """Flatten a name->tensor mapping (reference state_dict names) into the ABI's 12-per-block order."""
    return [named[k.format(l=l)] for l in range(L) for k in BLOCK_PARAM_ORDER]


def route_bucket(idx, R, G, cap, out=None, skip_row=-1):
    """Owner bucketing of lookups into a row-sharded table (re_route_bucket): -> (buckets int64 [G, cap] local row ids, -1 = unused;
    slot int64 [n]; counts int32 [G + 1], the last word = dropped lookups).  No host sync; fixed capacity per peer.
    skip_row >= 0: lookups of that row (the padding row) take no slot (slot -1) and are not counted as dropped."""
    _req(idx, torch.int64, "idx")
    n = idx.numel()
    dev = idx.device
    if out is None:
        out = (torch.empty((G, cap), dtype=torch.int64, device=dev), torch.empty(max(n, 1), dtype=torch.int64, device=dev)[:n],
               torch.empty(G + 1, dtype=torch.int32, device=dev))
    buckets, slot, counts = out
    L = lib.load()
    ws = _ws(L.re_route_workspace_bytes(n, G), dev)
    lib.check(L.re_route_bucket(_p(idx), n, int(R), int(G), int(cap), int(skip_row), _p(buckets), _p(slot) if n else None, _p(counts), _p(ws),
                                ws.numel(), _stream()), "re_route_bucket")
    return buckets, slot, counts


_NCU = {}


def num_cus(device=None):
    """Compute units of the device (the encoder kernels' launch grid: one resident workgroup per CU)."""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    if dev not in _NCU:
        _NCU[dev] = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    return _NCU[dev]


class PreparedBatch:
    """What `sasrec_batch_prep` leaves on the device for one (seq, pos, neg) batch: views into ONE uint8 blob."""
    __slots__ = ("B", "S", "blob", "seq", "pos", "neg", "rows_all", "valid", "count", "plan", "split", "weights_ready")


_PREP_LAYOUT, _PREP_VIEWS = {}, {}


def prep_layout(B, S):
    """Byte offsets of a prepared batch's arrays inside its blob (256-byte aligned).  -> (offsets {name: (off, nbytes)}, total)."""
    if (B, S) not in _PREP_LAYOUT:
        _PREP_LAYOUT[(B, S)] = _prep_layout(B, S)
    return _PREP_LAYOUT[(B, S)]


def _prep_layout(B, S):
    n, off, o = B * S, {}, 0
    for name, nbytes in (("seq", 8 * n), ("pos", 8 * n), ("neg", 8 * n), ("rows_all", 24 * n), ("valid", n), ("count", 4),
                         ("plan", int(lib.load().re_sasrec_plan_bytes(B, S)))):
        off[name] = (o, nbytes)
        o += (nbytes + 255) // 256 * 256
    return off, o


def prep_views(blob, B, S, cached=False):
    """The blob's arrays as views.  cached: a STATIC blob (a captured step's staging buffer) is cut once -- the ten view constructions
    were 20 us of the 43 us the staging call cost on the host per step."""
    if cached:
        key = (blob.data_ptr(), blob.numel(), B, S)
        pb = _PREP_VIEWS.get(key)
        if pb is None or pb.blob is not blob:
            if len(_PREP_VIEWS) >= 64:
                _PREP_VIEWS.clear()
            pb = _PREP_VIEWS[key] = prep_views(blob, B, S)
        return pb
    off, total = prep_layout(B, S)
    if blob.numel() < total:
        raise ValueError("recengine: prepared-batch blob too small")
    cut = lambda k, dt: blob[off[k][0]:off[k][0] + off[k][1]].view(dt)  # noqa: E731
    pb = PreparedBatch()
    pb.B, pb.S, pb.blob = B, S, blob
    pb.seq, pb.pos, pb.neg = (cut(k, torch.int64).view(B, S) for k in ("seq", "pos", "neg"))
    pb.rows_all, pb.valid, pb.count, pb.plan = cut("rows_all", torch.int64), cut("valid", torch.uint8), cut("count", torch.int32), cut("plan", torch.uint8)
    return pb


def tile_step_certain(B, S, D, split, tile, tile_wgs):
    """The step of this shape ALWAYS runs on the tile kernels when its plans are made with them forced (re_sasrec_tile_step_certain): the
    preparation calls then force them (_plan_flags) and the step leaves the workgroup-per-item launch out."""
    return bool(tile) and not split and int(tile_wgs) == 2 and int(D) == 64 and bool(lib.load().re_sasrec_tile_step_certain(int(B), int(S), int(D)))


def _plan_flags(split, tile, tile_wgs, B=0, S=0):
    """re_sasrec_batch_prep's `split_long` mask: & 1 split long sequences over two work items; & 2 never hand the step to the tile kernels;
    & 4 always; & 8 the tile kernels hold TWO workgroups per CU (D = 64: csrc/enc_common.h enc_tile_wg_per_cu) -- the plan's rule counts
    resident workgroups."""
    force = tile == "always" or (B and tile_step_certain(B, S, 64, split, tile, tile_wgs))      # (two per CU: D = 64)
    return int(bool(split)) | (0 if tile else 2) | (4 if force else 0) | (8 if tile_wgs == 2 else 0)


def sasrec_batch_prep(seq, pos=None, neg=None, blob=None, state=None, seed=0, step=1, lr=1e-3, beta1=0.9, beta2=0.999, max_tiles=4, split=False,
                      ncu=None, weights=None, loss_acc=None, tile=True, tile_wgs=1):
    """Batch preparation as ONE launch (re_sasrec_batch_prep): valid mask, count, scatter destination rows, the encoder's work plan;
    with `blob` (a static buffer of prep_layout(B, S) bytes) also copies (seq, pos, neg) into it and, with `state` (int32[4]),
    writes the step scalars -- the staging launch of a captured step.  -> PreparedBatch (views into the blob).
    split: sequences of 3 - 4 tiles may become two work items in two workgroups (training launches with a zero-initialised tape only).
    ncu: the number of workgroups the plan's items should fill (default: the device's CUs, one workgroup per CU; the wave-per-tile
    step at D = 64 takes 1024 -- one tile per item while the batch allows it).
    weights = (block_tensors, last_w, last_b, L, tape, ws) (D = 64): the launch also prepares the one-tile-per-workgroup step's weight
    fragments in `ws` (re_sasrec_batch_prep_w); the PreparedBatch then says `weights_ready` and sasrec_encoder_step skips its own.
    loss_acc = (prev_loss[1], acc[1], weight): acc += prev_loss * weight inside this launch (the previous step's loss into an epoch sum).
    tile=False: the plan never hands the training step to the one-tile-per-workgroup kernels (the workgroup-per-item kernels run it).
    (The tile kernels hold one workgroup per CU: more than 3/4 x ncu tiles of long sequences make the plan fall back as well.)"""
    _req(seq, torch.int64, "seq")
    B, S = seq.shape
    if pos is not None:
        _req(pos, torch.int64, "pos"); _req(neg, torch.int64, "neg")
    copy = blob is not None
    if blob is None:
        blob = torch.empty(prep_layout(B, S)[1], dtype=torch.uint8, device=seq.device)
    pb = prep_views(_req(blob, torch.uint8, "blob"), B, S, cached=copy)
    if state is not None:
        _req(state, torch.int32, "state")
    have = pos is not None
    args = (_p(seq), _p(pos), _p(neg), B, S, int(ncu) if ncu else num_cus(seq.device), int(max_tiles), _plan_flags(split, tile, tile_wgs, B, S),
            _p(pb.seq) if copy else None, _p(pb.pos) if copy and have else None, _p(pb.neg) if copy and have else None,
            _p(pb.valid) if have else None, _p(pb.count), _p(pb.rows_all) if have else None, _p(pb.plan), pb.plan.numel(),
            _p(state), int(seed) & 0xFFFFFFFF, int(step), float(lr), float(beta1), float(beta2))
    if weights is not None or loss_acc is not None:
        lib.check(lib.load().re_sasrec_batch_prep_w(*args, *_weight_args(weights), *_loss_args(loss_acc), _stream()), "re_sasrec_batch_prep_w")
    else:
        lib.check(lib.load().re_sasrec_batch_prep(*args, _stream()), "re_sasrec_batch_prep")
    pb.weights_ready = weights is not None
    if not copy:
        pb.seq, pb.pos, pb.neg = seq, pos, neg
    pb.split = bool(split)
    return pb


class MarshalledWeights:
    """`weights` of sasrec_batch_prep / sasrec_step_stage already turned into ABI arguments (the pointer table of the 12 L block tensors and the
    view constructions were ~50 us of host time per step): make it once per (engine, batch shape) with marshal_weights -- the tensors must stay
    where they are (parameter arenas do), and are kept alive here."""
    __slots__ = ("args", "keep")


def marshal_weights(weights):
    if weights is None or isinstance(weights, MarshalledWeights):
        return weights
    m = MarshalledWeights()
    m.args, m.keep = _weight_args(weights), weights
    return m


def _weight_args(weights):
    if weights is None:
        return (None, None, None, 0, 0, None, 0, None, 0)
    if isinstance(weights, MarshalledWeights):
        return weights.args
    bt, lw, lb, L, tape, ws = weights
    return (_ptr_table(bt), _p(lw), _p(lb), int(L), int(lw.numel()), _p(tape), tape.numel() * 4, _p(ws), ws.numel())


def _loss_args(loss_acc):
    if loss_acc is None:
        return (None, None, 0.0)
    prev, acc, w = loss_acc
    _req(prev, torch.float32, "prev_loss"); _req(acc, torch.float32, "loss_acc")
    return (_p(prev), _p(acc), float(w))


def sasrec_sample_prep(inter, order, b0, B, S, sample_seed, sample_step, blob, state=None, seed=0, step=1, lr=1e-3, beta1=0.9, beta2=0.999,
                       max_tiles=4, split=False, ncu=None, weights=None, users=None, loss_acc=None, tile=True, tile_wgs=1):
    """SAMPLE + PREPARE as one launch (re_seq_train_sample_prep): rows b0 .. b0 + B of the epoch's user order `order`, sampled as
    recboard_amd.sampler.seq_train_sample would, written straight into the staging `blob` with everything sasrec_batch_prep derives.
    inter: recboard_amd.sampler.DeviceInteractions.  -> PreparedBatch (views into the blob)."""
    pb = prep_views(_req(blob, torch.uint8, "blob"), B, S, cached=True)
    if state is not None:
        _req(state, torch.int32, "state")
    wargs = _weight_args(weights) + _loss_args(loss_acc)
    lib.check(lib.load().re_seq_train_sample_prep(_p(inter.ptr), _p(inter.items), _p(inter.sorted), _p(order), order.numel(), int(b0), inter.num_items,
                                                  int(sample_seed) & 0xFFFFFFFF, int(sample_step) & 0xFFFFFFFF, _p(users), B, S,
                                                  int(ncu) if ncu else num_cus(blob.device), int(max_tiles), _plan_flags(split, tile, tile_wgs, B, S), _p(pb.seq), _p(pb.pos),
                                                  _p(pb.neg), _p(pb.valid), _p(pb.count), _p(pb.rows_all), _p(pb.plan), pb.plan.numel(), _p(state),
                                                  int(seed) & 0xFFFFFFFF, int(step), float(lr), float(beta1), float(beta2), *wargs, _stream()),
              "re_seq_train_sample_prep")
    pb.split = bool(split)
    pb.weights_ready = weights is not None
    return pb


def max_tiles(D):
    """Tiles of 16 rows a work item of the encoder kernels holds in LDS (what the plan groups short sequences by)."""
    return 4 if D == 64 else 2


def sasrec_plan(seq, D=64):
    """The encoder's work plan alone (evaluation / unit tests).  -> opaque uint8 tensor."""
    return sasrec_batch_prep(seq, max_tiles=max_tiles(D)).plan


def sasrec_encoder_fwd(x0, seq, block_tensors, last_w, last_b, L, drop_p=0.0, seed=0, need_tape=False, out=None, tape=None,
                       plan=None, seed_dev=None, embed=None):
    """u = lastLN(blocks(x0)) fused (re_sasrec_encoder_fwd).  embed = (E, P, scale): x0 is built inside the kernel (x0 = None).
    -> (u [B,S,D], tape or None)."""
    _req(seq, torch.int64, "seq")
    B, S = seq.shape
    if embed is not None:
        E, Ptab, scale = embed
        _req(E, torch.float32, "E"); _req(Ptab, torch.float32, "P")
        D, x0 = E.shape[1], None
    else:
        _req(x0, torch.float32, "x0")
        E, Ptab, scale, D = None, None, 0.0, x0.shape[2]
    Lb = lib.load()
    u = out if out is not None else torch.empty((B, S, D), dtype=torch.float32, device=seq.device)
    fill_pads = 0 if (need_tape or tape is not None) else 1      # inference: pad positions read lastLN.bias; training does not write them
    if tape is None and (need_tape or (D == 128 and S > 16 * max_tiles(D))):
        # (D = 128, S > 32: the parts of a long sequence hand k, v over through the tape, inference included)
        tape = torch.zeros(Lb.re_sasrec_tape_bytes(B, S, D, L) // 4, dtype=torch.float32, device=seq.device)
    if plan is None:
        plan = sasrec_plan(seq, D)
    tbl = _ptr_table(block_tensors)
    lib.check(Lb.re_sasrec_encoder_fwd(_p(x0), _p(E), 0 if E is None else E.shape[0], _p(Ptab), float(scale), _p(seq), B, S, D, L, tbl,
                                       _p(last_w), _p(last_b), float(drop_p), int(seed) & 0xFFFFFFFF, _p(seed_dev), _p(plan),
                                       num_cus(seq.device), _p(u), _p(tape), 0 if tape is None else tape.numel() * 4, fill_pads,
                                       _stream()), "re_sasrec_encoder_fwd")
    return u, (tape if need_tape or fill_pads == 0 else None)


def sasrec_embed_encoder_fwd(E, P, seq, scale, block_tensors, last_w, last_b, L, drop_p=0.0, seed=0, need_tape=False, out=None,
                             tape=None, plan=None, seed_dev=None):
    """sasrec_embed + sasrec_encoder_fwd in one launch.  -> (u [B,S,D], tape or None)."""
    return sasrec_encoder_fwd(None, seq, block_tensors, last_w, last_b, L, drop_p, seed, need_tape, out, tape, plan, seed_dev,
                              embed=(E, P, scale))


def sasrec_encoder_bwd(dU, seq, block_tensors, last_w, last_b, L, drop_p, seed, tape, block_grads, g_last_w, g_last_b,
                       out=None, ws=None, plan=None, seed_dev=None, embed_scale=None, dP=None, dU_rows=None, out_rows=None):
    """-> dx0 [B,S,D]; OVERWRITES the tensors in block_grads / g_last_* with the parameter gradients.  With dP (and embed_scale)
    re_sasrec_embed_bwd is fused in: the result is the item-gradient contribution rows and dP [S,D] the position-table gradient.
    dU_rows [plan rows, D] (sasrec_loss_rows) replaces dU (pass dU=None and `out`); out_rows receives dx0's rows in compact order."""
    _req(seq, torch.int64, "seq"); _req(tape, torch.float32, "tape")
    if dU_rows is not None:
        _req(dU_rows, torch.float32, "dU_rows")
        if out is None:
            raise ValueError("recengine: dU_rows needs `out`")
        B, S, D = out.shape
        dev = dU_rows.device
    else:
        _req(dU, torch.float32, "dU")
        B, S, D = dU.shape
        dev = dU.device
    if out_rows is not None:
        _req(out_rows, torch.float32, "out_rows")
    Lb = lib.load()
    dx0 = out if out is not None else torch.empty_like(dU)
    _req(dx0, torch.float32, "out")
    if dP is not None:
        _req(dP, torch.float32, "dP")
    if ws is None:
        ws = _ws(Lb.re_sasrec_encoder_bwd_workspace_bytes(B, S, D, L), dev)
    if plan is None:
        plan = sasrec_plan(seq, D)
    tp, tg = _ptr_table(block_tensors), _ptr_table(block_grads)
    lib.check(Lb.re_sasrec_encoder_bwd(_p(dU), _p(seq), B, S, D, L, tp, _p(last_w), _p(last_b), float(drop_p),
                                       int(seed) & 0xFFFFFFFF, _p(seed_dev), _p(tape), _p(plan), num_cus(dev),
                                       float(embed_scale or 0.0), _p(dx0), _p(dP), tg, _p(g_last_w), _p(g_last_b), _p(dU_rows),
                                       _p(out_rows), _p(ws), ws.numel(), _stream()), "re_sasrec_encoder_bwd")
    return dx0


def sasrec_encoder_fwd_loss(E, Ptab, seq, pos, neg, scale, block_tensors, last_w, last_b, L, drop_p, seed, plan, kind, count, u, tape,
                            dU_rows, g_rows, keys, ws, e_off=1, loss=None, seed_dev=None):
    """Training forward + pair criterion in one launch (re_sasrec_encoder_fwd_loss): fills u [B,S,D], tape, dU_rows, g_rows[1:3], keys;
    -> loss[1].  ws: 256 zeroed bytes (kept zero by the kernel)."""
    _req(E, torch.float32, "E"); _req(Ptab, torch.float32, "Ptab"); _req(seq, torch.int64, "seq"); _req(pos, torch.int64, "pos")
    _req(neg, torch.int64, "neg"); _req(u, torch.float32, "u"); _req(tape, torch.float32, "tape"); _req(dU_rows, torch.float32, "dU_rows")
    _req(g_rows, torch.float32, "g_rows"); _req(keys, torch.int32, "keys"); _req(count, torch.int32, "count"); _req(ws, torch.uint8, "ws")
    B, S = seq.shape
    R, D = E.shape
    NR = sasrec_plan_rows(B, S)
    if dU_rows.numel() != NR * D or g_rows.numel() != 3 * NR * D or keys.numel() != 3 * NR or u.numel() != B * S * D:
        raise ValueError("recengine: sasrec_encoder_fwd_loss buffer shapes")
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=E.device)
    tp = _ptr_table(block_tensors)
    lib.check(lib.load().re_sasrec_encoder_fwd_loss(_p(E), R, _p(Ptab), float(scale), _p(seq), _p(pos), _p(neg), B, S, D, L, tp, _p(last_w),
                                                    _p(last_b), float(drop_p), int(seed) & 0xFFFFFFFF, _p(seed_dev), _p(plan),
                                                    num_cus(E.device), _p(u), _p(tape), tape.numel() * 4, int(e_off), int(kind), _p(count),
                                                    _p(loss), _p(dU_rows), _p(g_rows), _p(keys), _p(ws), ws.numel(), _stream()),
              "re_sasrec_encoder_fwd_loss")
    return loss


def sasrec_encoder_step(E, Ptab, seq, pos, neg, scale, block_tensors, last_w, last_b, L, drop_p, seed, plan, kind, count, u, tape,
                        dU_rows, g_rows, keys, loss_ws, dx0, dP, block_grads, g_last_w, g_last_b, ws, e_off=1, loss=None, seed_dev=None, part=0,
                        adam=None):
    """Forward + criterion + backward of the encoder per work item in ONE launch, then the weight gradients (re_sasrec_encoder_step):
    what sasrec_encoder_fwd_loss + sasrec_encoder_bwd(dU_rows=..., out_rows=g_rows[0], dP=...) compute, bit for bit.  -> loss[1].
    part (re_sasrec_encoder_step_part): a mask -- 1 tile kernels, 2 workgroup-per-item kernel, 4 weight gradients, + 8 weights prepared.
    adam (AdamFuse over the arenas the gradient tensors are views of): the reduction that finishes the encoder's gradients applies their
    dense Adam update too."""
    _req(E, torch.float32, "E"); _req(Ptab, torch.float32, "Ptab"); _req(seq, torch.int64, "seq"); _req(pos, torch.int64, "pos")
    _req(neg, torch.int64, "neg"); _req(u, torch.float32, "u"); _req(tape, torch.float32, "tape"); _req(dU_rows, torch.float32, "dU_rows")
    _req(g_rows, torch.float32, "g_rows"); _req(keys, torch.int32, "keys"); _req(count, torch.int32, "count"); _req(loss_ws, torch.uint8, "loss_ws")
    _req(dx0, torch.float32, "dx0"); _req(dP, torch.float32, "dP"); _req(ws, torch.uint8, "ws")
    B, S = seq.shape
    R, D = E.shape
    NR = sasrec_plan_rows(B, S)
    if dU_rows.numel() != NR * D or g_rows.numel() != 3 * NR * D or keys.numel() != 3 * NR or u.numel() != B * S * D or dx0.numel() != B * S * D:
        raise ValueError("recengine: sasrec_encoder_step buffer shapes")
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=E.device)
    tp, tg = _ptr_table(block_tensors), _ptr_table(block_grads)
    lib.check(lib.load().re_sasrec_encoder_step_part(_p(E), R, _p(Ptab), float(scale), _p(seq), _p(pos), _p(neg), B, S, D, L, tp, _p(last_w),
                                                     _p(last_b), float(drop_p), int(seed) & 0xFFFFFFFF, _p(seed_dev), _p(plan), num_cus(E.device),
                                                     _p(u), _p(tape), tape.numel() * 4, int(e_off), int(kind), _p(count), _p(loss), _p(dU_rows),
                                                     _p(g_rows), _p(keys), _p(loss_ws), loss_ws.numel(), _p(dx0), _p(dP), tg, _p(g_last_w),
                                                     _p(g_last_b), _p(ws), ws.numel(), int(part), ctypes.byref(adam) if adam is not None else None,
                                                     _stream()), "re_sasrec_encoder_step_part")
    return loss


def sasrec_tape_errors(tape, B, S):
    """Number of hand-over time-outs recorded in a tape's flag words (split long sequences; 0 in a healthy run).  Host sync."""
    mt = B * ((S + 15) // 16)
    return int(tape[-16:].view(torch.int32)[0].item()) if tape.numel() >= mt * 8 + 16 else 0


def sasrec_tape_reset_flags(tape, B, S):
    """Zero a tape's hand-over flags and error word: after a time-out (or a rejected split plan) a producer's late flag store would
    otherwise be read by the next replay as "published"."""
    mt = B * ((S + 15) // 16)
    if tape.numel() >= mt * 8 + 16:
        tape[-(mt * 8 + 16):].zero_()


TAPE_FIELDS = ("per_block", "X", "A", "Q", "K", "V", "O", "X1", "Y", "HR", "P", "SA", "SF", "PP", "MK", "XL", "SL", "FLAGS", "total")


def sasrec_tape_array(tape, plan, B, S, D, L, name, block):
    """One [NR, D] activation array of a training tape, scattered back to [B, S, D] through the plan's row map (rows without a
    compact row -- the pads in front of a sequence -- stay zero).  Diagnostics and tests (host syncs)."""
    import ctypes
    out = (ctypes.c_int64 * len(TAPE_FIELDS))()
    lib.check(lib.load().re_sasrec_tape_layout(B, S, D, L, ctypes.cast(out, ctypes.c_void_p), len(TAPE_FIELDS)), "re_sasrec_tape_layout")
    off = dict(zip(TAPE_FIELDS, out))
    w = plan.view(torch.int32)
    mt = B * ((S + 15) // 16)
    nr = 16 * int(w[1])
    rm = w[(8 + mt + 1) // 2 * 2:][: 2 * 16 * mt].view(-1, 2)[:nr, 0].long()          # gid of every compact row (-1: dummy)
    a = tape[block * off["per_block"] + off[name]:][: nr * D].view(nr, D)
    dense = torch.zeros(B * S, D, dtype=tape.dtype, device=tape.device)
    live = rm >= 0
    dense[rm[live]] = a[live]
    return dense.view(B, S, D)


def sasrec_plan_rows(B, S):
    """Upper bound of the number of compact rows of a batch plan (re_sasrec_plan_rows)."""
    return int(lib.load().re_sasrec_plan_rows(int(B), int(S)))


def sasrec_loss_rows(U, E, seq, pos, neg, plan, kind, count, dU_rows, g_rows, keys, e_off=1, loss=None, ws=None):
    """Pair criteria on the plan's compact rows (re_sasrec_loss_rows) -> loss[1]; fills dU_rows [NR,D], g_rows[1:3] ([3,NR,D]) and
    keys int32 [3,NR].  ws: zero-filled uint8 tensor of re_sasrec_loss_rows_workspace_bytes() (kept zero by the kernel)."""
    _req(U, torch.float32, "U"); _req(E, torch.float32, "E"); _req(seq, torch.int64, "seq"); _req(pos, torch.int64, "pos")
    _req(neg, torch.int64, "neg"); _req(dU_rows, torch.float32, "dU_rows"); _req(g_rows, torch.float32, "g_rows")
    _req(keys, torch.int32, "keys"); _req(count, torch.int32, "count")
    B, S = seq.shape
    R, D = E.shape
    NR = sasrec_plan_rows(B, S)
    if dU_rows.numel() != NR * D or g_rows.numel() != 3 * NR * D or keys.numel() != 3 * NR or U.numel() != B * S * D:
        raise ValueError("recengine: sasrec_loss_rows buffer shapes")
    Lb = lib.load()
    if ws is None:
        ws = torch.zeros(Lb.re_sasrec_loss_rows_workspace_bytes(), dtype=torch.uint8, device=U.device)
    if loss is None:
        loss = torch.empty(1, dtype=torch.float32, device=U.device)
    lib.check(Lb.re_sasrec_loss_rows(_p(U), _p(E), R, D, int(e_off), _p(seq), _p(pos), _p(neg), B, S, _p(plan), int(kind), _p(count),
                                     _p(loss), _p(dU_rows), _p(g_rows), _p(keys), _p(ws), ws.numel(), _stream()), "re_sasrec_loss_rows")
    return loss


class AdamFuse(ctypes.Structure):
    """re_adam_fuse (include/recengine.h): the dense Adam of the gradients a launch finishes, applied by that launch."""
    _fields_ = [("grad_base", ctypes.c_void_p), ("param", ctypes.c_void_p), ("m", ctypes.c_void_p), ("v", ctypes.c_void_p),
                ("hyper", ctypes.c_void_p), ("beta1", ctypes.c_double), ("beta2", ctypes.c_double), ("eps", ctypes.c_double),
                ("weight_decay", ctypes.c_double)]


def adam_fuse(grad, param, m, v, hyper, beta1, beta2, eps, weight_decay):
    """-> an AdamFuse over arenas of one layout (keep the tensors alive while it is in use)."""
    for t, nm in ((grad, "grad"), (param, "param"), (m, "m"), (v, "v"), (hyper, "hyper")):
        _req(t, torch.float32, nm)
    return AdamFuse(_note(grad), _note(param), _note(m), _note(v), _note(hyper), float(beta1), float(beta2), float(eps),
                    float(weight_decay))


def scatter_add_rows_small(g, keys, R, out, n_regions=1, region_stride=None, n_dev=None, n_mul=1, n=None, padding_idx=0, scale=1.0, adam=None):
    """Dense [R, D] sum of contribution rows by int32 destination keys without a sort (re_scatter_add_rows_small): `out` is fully
    overwritten.  n keys per region: `n` (host) or n_dev[0] * n_mul (device int32).
    adam (AdamFuse over the table's param / m / v rows [0, R)): the table's dense Adam update is applied by the same launch
    (re_scatter_adam_rows_small); `out` may then be None (the gradient is not written)."""
    _req(g, torch.float32, "g"); _req(keys, torch.int32, "keys")
    if out is not None or adam is None:
        _req(out, torch.float32, "out")
    D = g.shape[-1]
    if region_stride is None:
        region_stride = keys.numel() // n_regions
    if n is None and n_dev is None:
        n = region_stride
    if n_dev is not None:
        _req(n_dev, torch.int32, "n_dev")
    if (out is not None and out.numel() != R * D) or g.numel() < n_regions * region_stride * D:
        raise ValueError("recengine: scatter_add_rows_small buffer shapes")
    if adam is not None:
        lib.check(lib.load().re_scatter_adam_rows_small(_p(g), _p(keys), int(n_regions), int(region_stride), _p(n_dev), int(n_mul), int(n or 0), D,
                                                        int(R), int(padding_idx), float(scale), _p(out), ctypes.byref(adam), _stream()),
                  "re_scatter_adam_rows_small")
        return out
    lib.check(lib.load().re_scatter_add_rows_small(_p(g), _p(keys), int(n_regions), int(region_stride), _p(n_dev), int(n_mul),
                                                   int(n or 0), D, int(R), int(padding_idx), float(scale), _p(out), _stream()),
              "re_scatter_add_rows_small")
    return out


class NextPrep(ctypes.Structure):
    """re_next_prep (include/recengine.h): the next batch's preparation as jobs of a step's tail launch."""
    _fields_ = [("mail", ctypes.c_void_p), ("B", ctypes.c_int64), ("S", ctypes.c_int64), ("ncu", ctypes.c_int32), ("max_tiles", ctypes.c_int32),
                ("split_long", ctypes.c_int32), ("seq_out", ctypes.c_void_p), ("pos_out", ctypes.c_void_p), ("neg_out", ctypes.c_void_p),
                ("valid", ctypes.c_void_p), ("count", ctypes.c_void_p), ("rows_all", ctypes.c_void_p), ("plan", ctypes.c_void_p),
                ("plan_bytes", ctypes.c_size_t)]


def next_prep(mail, blob, B, S, max_tiles=4, split=False, ncu=None, tile=True, tile_wgs=1):
    """-> NextPrep: the batch whose addresses sasrec_step_stage leaves in `mail` (int64[4], device) is prepared into the staging `blob`
    (sasrec_batch_prep(..., blob=blob)'s outputs) by the tail launch this is handed to.  Keeps `mail` and `blob` alive."""
    _req(mail, torch.int64, "mail"); _req(blob, torch.uint8, "blob")
    pb = prep_views(blob, B, S, cached=True)
    n = NextPrep(_p(mail), B, S, int(ncu) if ncu else num_cus(blob.device), int(max_tiles), _plan_flags(split, tile, tile_wgs, B, S), _p(pb.seq), _p(pb.pos),
                 _p(pb.neg), _p(pb.valid), _p(pb.count), _p(pb.rows_all), _p(pb.plan), pb.plan.numel())
    n._keep = (mail, blob, pb)
    return n


MAIL_WORDS = 16      # int64 words of a mailbox (RE_MAIL_BYTES)


def sasrec_step_stage(state, seed, step, lr, beta1, beta2, B, S, mail=None, next_batch=None, weights=None, loss_acc=None, next_ticket=None):
    """The launch in front of a captured step whose batch the previous step's tail launch prepared (re_sasrec_step_stage): the step scalars,
    the loss fold, the tile kernels' weight fragments, and -- for this step's tail launch -- where the FOLLOWING batch comes from: its tensors
    (next_batch) or its sampling source (next_ticket: a recboard_amd.sampler.SampleTicket; re_sasrec_step_stage_sample); neither: none."""
    _req(state, torch.int32, "state")
    if mail is not None and mail.numel() < MAIL_WORDS:
        raise ValueError("recengine: a mailbox is MAIL_WORDS int64 words")
    if next_ticket is not None:
        t = next_ticket
        if (t.B, t.S) != (B, S):
            raise ValueError("recengine: the next ticket must have the captured step's shape")
        _req(mail, torch.int64, "mail")
        lib.check(lib.load().re_sasrec_step_stage_sample(_p(state), int(seed) & 0xFFFFFFFF, int(step), float(lr), float(beta1), float(beta2), _p(mail),
                                                         _p(t.inter.ptr), _p(t.inter.items), _p(t.inter.sorted), _p(t.order), t.order.numel(), int(t.b0),
                                                         t.inter.num_items, int(t.seed) & 0xFFFFFFFF, int(t.step) & 0xFFFFFFFF, _p(t.users), B, S,
                                                         *_weight_args(weights), *_loss_args(loss_acc), _stream()), "re_sasrec_step_stage_sample")
        return
    ns = npos = nn = None
    if next_batch is not None:
        ns, npos, nn = next_batch
        _req(ns, torch.int64, "next seq"); _req(npos, torch.int64, "next pos"); _req(nn, torch.int64, "next neg")
        if tuple(ns.shape) != (B, S) or tuple(npos.shape) != (B, S) or tuple(nn.shape) != (B, S):
            raise ValueError("recengine: the next batch must have the captured step's shape")
    if mail is not None:
        _req(mail, torch.int64, "mail")
    lib.check(lib.load().re_sasrec_step_stage(_p(state), int(seed) & 0xFFFFFFFF, int(step), float(lr), float(beta1), float(beta2), _p(mail), _p(ns),
                                              _p(npos), _p(nn), B, S, *_weight_args(weights), *_loss_args(loss_acc), _stream()), "re_sasrec_step_stage")


def sasrec_step_tail(g_rows, keys, R, out, n_dev, n_mul, seq, L, plan, tape, dx0, scale, dP, block_grads, g_last_w, g_last_b, ws, ticket,
                     table_adam=None, enc_adam=None, n_regions=3, padding_idx=0, next=None):
    """scatter_add_rows_small(g_rows, keys, R, out, n_regions, n_dev=, n_mul=, adam=table_adam) and sasrec_encoder_step(part=4, adam=enc_adam)
    as ONE launch + the reduction (re_sasrec_step_tail; D = 64): the scatter-add's workgroups take the weight-gradient jobs when their rows are
    done.  Bit-identical to the two calls.  ticket: a zero uint32/int32[1] of the caller's (left zero).  next (NextPrep): the launch also
    prepares the next batch (the one sasrec_step_stage named) into the other captured copy's staging buffers."""
    _req(g_rows, torch.float32, "g_rows"); _req(keys, torch.int32, "keys"); _req(n_dev, torch.int32, "n_dev"); _req(seq, torch.int64, "seq")
    _req(tape, torch.float32, "tape"); _req(dx0, torch.float32, "dx0"); _req(dP, torch.float32, "dP"); _req(ws, torch.uint8, "ws")
    _req(ticket, torch.int32, "ticket")
    if out is not None or table_adam is None:
        _req(out, torch.float32, "out")
    B, S = seq.shape
    D = g_rows.shape[-1]
    region_stride = keys.numel() // n_regions
    if (out is not None and out.numel() != R * D) or g_rows.numel() < n_regions * region_stride * D or dx0.numel() != B * S * D:
        raise ValueError("recengine: sasrec_step_tail buffer shapes")
    tg = _ptr_table(block_grads)
    lib.check(lib.load().re_sasrec_step_tail(_p(g_rows), _p(keys), int(n_regions), int(region_stride), _p(n_dev), int(n_mul), int(R), int(padding_idx),
                                             _p(out), ctypes.byref(table_adam) if table_adam is not None else None, _p(seq), B, S, D, int(L), _p(plan),
                                             num_cus(seq.device), _p(tape), tape.numel() * 4, _p(dx0), float(scale), _p(dP), tg, _p(g_last_w),
                                             _p(g_last_b), _p(ws), ws.numel(), ctypes.byref(enc_adam) if enc_adam is not None else None,
                                             _p(ticket), ctypes.byref(next) if next is not None else None, _stream()), "re_sasrec_step_tail")
    return out


def sasrec_step_tail_sparse(g_rows, keys, W, m, v, hyper, beta1, beta2, eps, weight_decay, n_dev, n_mul, seq, L, plan, tape, dx0, scale, dP,
                            block_grads, g_last_w, g_last_b, ws, ticket, enc_adam=None, padding_idx=0, next=None):
    """sparse_adam_rows_small(g_rows, keys [regions, stride] int32, W, m, v, hyper=, n_dev=, n_mul=) and sasrec_encoder_step(part=4,
    adam=enc_adam) as ONE launch + the reduction (re_sasrec_step_tail_sparse): the tail of a large-table step.  Bit-identical to the two calls."""
    _req(g_rows, torch.float32, "g_rows"); _req(keys, torch.int32, "keys"); _req(n_dev, torch.int32, "n_dev"); _req(seq, torch.int64, "seq")
    _req(tape, torch.float32, "tape"); _req(dx0, torch.float32, "dx0"); _req(dP, torch.float32, "dP"); _req(ws, torch.uint8, "ws")
    _req(ticket, torch.int32, "ticket"); _req(hyper, torch.float32, "hyper")
    for t, nme in ((W, "W"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nme)
    B, S = seq.shape
    R, D = W.shape
    regions, stride = keys.shape
    if g_rows.numel() != keys.numel() * D or dx0.numel() != B * S * D:
        raise ValueError("recengine: sasrec_step_tail_sparse buffer shapes")
    tg = _ptr_table(block_grads)
    lib.check(lib.load().re_sasrec_step_tail_sparse(_p(g_rows), _p(keys), int(regions), int(stride), _p(n_dev), int(n_mul), R, int(padding_idx), _p(W),
                                                    _p(m), _p(v), _p(hyper), float(beta1), float(beta2), float(eps), float(weight_decay), _p(seq), B, S, D,
                                                    int(L), _p(plan), num_cus(seq.device), _p(tape), tape.numel() * 4, _p(dx0), float(scale), _p(dP), tg,
                                                    _p(g_last_w), _p(g_last_b), _p(ws), ws.numel(),
                                                    ctypes.byref(enc_adam) if enc_adam is not None else None, _p(ticket),
                                                    ctypes.byref(next) if next is not None else None, _stream()),
              "re_sasrec_step_tail_sparse")


def sasrec_encoder_embed_bwd(dU, seq, scale, block_tensors, last_w, last_b, L, drop_p, seed, tape, block_grads, g_last_w, g_last_b, dP,
                             out=None, ws=None, plan=None, seed_dev=None):
    """sasrec_encoder_bwd + sasrec_embed_bwd in one pass: -> item-gradient contribution rows [B,S,D]; OVERWRITES block_grads /
    g_last_* / dP with the parameter gradients."""
    return sasrec_encoder_bwd(dU, seq, block_tensors, last_w, last_b, L, drop_p, seed, tape, block_grads, g_last_w, g_last_b, out, ws,
                              plan, seed_dev, embed_scale=scale, dP=dP)


# ------------------------------------------------------------------------------------------------ K8
SPMM_LONG, SPMM_CHUNK = 512, 2048


class SpmmPlan:
    """Once per adjacency (the only host-visible preprocessing): rows in descending-degree order; the rows with more than
    SPMM_LONG non-zeros are cut into chunks of SPMM_CHUNK non-zeros that get a workgroup each."""

    def __init__(self, crow: torch.Tensor, D: int, split_row: int = 0, nt: bool = False):
        """split_row (0: none): rows [0, split_row) and [split_row, n) are two classes that gather from different parts of X (a bipartite
        adjacency: split_row = number of users); the short rows of each class are then walked by XCDs of their own (re_spmm_csr_split):
        each XCD's L2 holds the hot rows of ONE part of X.  The 8 XCD labels are shared out by the classes' gather cost -- non-zeros, weighted
        by the size of the part they gather from (the larger part misses more).  Yelp2018 shapes: 127.4 -> 118.1 us per propagation at 4 : 4
        (5 : 3: 120.2; 3 : 5: 142.8; profiles/r6_spmm_split.txt).
        nt: the CSR stream and the output rows with the non-temporal hint -- measured SLOWER (169.5 us: more L2 hits, and yet the stream's
        own loads get slower); kept as a switch for the record, off."""
        deg = crow[1:] - crow[:-1]
        self.row_order = torch.argsort(deg, descending=True, stable=True).contiguous()
        self.nlong = int((deg > SPMM_LONG).sum())
        self.split, self.xcd_share, self.flags = 0, 4, int(bool(nt))
        n = deg.numel()
        if 0 < int(split_row) < n:
            short = self.row_order[self.nlong:]
            c0, c1 = short[short < split_row], short[short >= split_row]             # (boolean masks keep the descending-degree order)
            if c0.numel() and c1.numel():
                w0, w1 = float(deg[c0].sum()) * (1.0 + (n - split_row) / n), float(deg[c1].sum()) * (1.0 + split_row / n)
                self.row_order = torch.cat([self.row_order[: self.nlong], c0, c1]).contiguous()
                self.split = self.nlong + int(c0.numel())
                self.xcd_share = min(7, max(1, int(round(8.0 * w0 / max(w0 + w1, 1.0)))))
        ldeg = deg[self.row_order[: self.nlong]]
        nch = (ldeg + SPMM_CHUNK - 1) // SPMM_CHUNK
        self.chunk_ptr = torch.zeros(self.nlong + 1, dtype=torch.int64, device=crow.device)
        self.chunk_ptr[1:] = torch.cumsum(nch, 0)
        self.nchunks = int(self.chunk_ptr[-1]) if self.nlong else 0
        self.chunk_row = torch.repeat_interleave(torch.arange(self.nlong, device=crow.device, dtype=torch.int32), nch).contiguous()
        # the rows' (first, end) positions in walking order: read beside the row id instead of behind it (re_spmm_csr_masked: row_ptrs)
        self.row_ptrs = torch.stack([crow[self.row_order], crow[self.row_order + 1]], 1).contiguous() if self.nlong else None
        # workspace: the long rows' chunk partials [nchunks, D] | (16-byte aligned) one int32 per long row -- how many of its chunks have arrived
        # in the running launch (flags & 2: the workgroup with a row's last chunk adds the partials inside the launch; zero between launches)
        # (flags & 2 is OFF: measured SLOWER on the Yelp2018 shape -- 130 us per propagation against 109: 459 release fences, each a write-back of
        #  an XCD's whole L2, in the middle of the row walk cost far more than the 6 us launch they replace; the switch stays for the record)
        self.in_launch = False
        if self.in_launch:
            self.flags |= 2
        self.ws = torch.zeros(self._ws_floats(D), dtype=torch.float32, device=crow.device)

    def _ws_floats(self, D):
        return max((self.nchunks * D + 3) // 4 * 4 + self.nlong, 4)


def spmm_plan(crow: torch.Tensor, D: int = 64, split_row: int = 0, nt: bool = False):
    return SpmmPlan(crow, D, split_row, nt)


def row_mask(rows, nbits, out=None):
    """-> int32 words, bit i set iff i occurs in rows (re_row_mask): which rows of a scatter's dense output can be non-zero."""
    _req(rows, torch.int64, "rows")
    out = out if out is not None else torch.empty((int(nbits) + 31) // 32, dtype=torch.int32, device=rows.device)
    _req(out, torch.int32, "out")
    lib.check(lib.load().re_row_mask(_p(rows), rows.numel(), int(nbits), _p(out), _stream()), "re_row_mask")
    return out


def spmm_csr(crow, col, val, plan, X, out, Z=None, beta=0.0, acc=None, acc_scale=0.0, acc_init=False, src_mask=None, z_mask=None):
    """out = A @ X (+ beta * Z); acc += acc_scale * out  (re_spmm_csr).  plan = spmm_plan(crow, D).
    acc_init (square adjacency): acc = acc_scale * (X + out) instead -- the running sum starts with this product's own input.
    src_mask (row_mask): a bit per row of X, 0 = the row is all zeros and is skipped.  z_mask: the same for Z's rows (where both are given
    they are ONE mask: Z and X the same kind of scatter)."""
    for t, nme in ((crow, "crow"), (col, "col")):
        _req(t, torch.int64, nme)
    for t, nme in ((val, "val"), (X, "X"), (out, "out")):
        _req(t, torch.float32, nme)
    nrows = crow.numel() - 1
    D = X.shape[1]
    if plan.ws.numel() < plan._ws_floats(D):
        plan.ws = torch.zeros(plan._ws_floats(D), dtype=torch.float32, device=X.device)
    flags = int(getattr(plan, "flags", 0)) | (4 if acc_init and acc is not None else 0)
    row_ptrs = getattr(plan, "row_ptrs", None)
    if z_mask is not None and Z is not None:
        if src_mask is not None and src_mask.data_ptr() != z_mask.data_ptr():
            raise ValueError("recengine: src_mask and z_mask must be the same mask")
        flags |= 8 | (16 if src_mask is None else 0)
        src_mask = z_mask
    if src_mask is not None or row_ptrs is not None:     # (src_mask: rows of X with a zero bit are all zeros: not fetched; same bits as without)
        if src_mask is not None:
            _req(src_mask, torch.int32, "src_mask")
            if src_mask.numel() * 32 < X.shape[0]:
                raise ValueError("recengine: src_mask needs a bit per row of X")
        lib.check(lib.load().re_spmm_csr_masked(_p(crow), _p(col), _p(val), nrows, X.shape[0], _p(plan.row_order), plan.nlong, int(plan.split),
                                                int(plan.xcd_share), flags, _p(plan.chunk_row), _p(plan.chunk_ptr), plan.nchunks, _p(X), D,
                                                _p(out), _p(Z), float(beta), _p(acc), float(acc_scale), _p(src_mask), _p(row_ptrs), _p(plan.ws),
                                                plan.ws.numel() * 4, _stream()), "re_spmm_csr_masked")
        return out
    if getattr(plan, "split", 0) or flags:
        lib.check(lib.load().re_spmm_csr_split(_p(crow), _p(col), _p(val), nrows, X.shape[0], _p(plan.row_order), plan.nlong, int(plan.split),
                                               int(plan.xcd_share), flags, _p(plan.chunk_row), _p(plan.chunk_ptr), plan.nchunks, _p(X), D,
                                               _p(out), _p(Z), float(beta), _p(acc), float(acc_scale), _p(plan.ws), plan.ws.numel() * 4, _stream()),
                  "re_spmm_csr_split")
        return out
    lib.check(lib.load().re_spmm_csr(_p(crow), _p(col), _p(val), nrows, X.shape[0], _p(plan.row_order), plan.nlong,
                                     _p(plan.chunk_row), _p(plan.chunk_ptr), plan.nchunks, _p(X), D, _p(out), _p(Z), float(beta),
                                     _p(acc), float(acc_scale), _p(plan.ws), plan.ws.numel() * 4, _stream()), "re_spmm_csr")
    return out


def rows_sqnorm(W, idx, scale, out, accumulate=False, ws=None):
    """out[0] (+)= scale * sum_i ||W[idx[i]]||^2  (re_rows_sqnorm)."""
    _req(W, torch.float32, "W"); _req(idx, torch.int64, "idx"); _req(out, torch.float32, "out")
    L = lib.load()
    if ws is None:
        ws = _ws(L.re_rows_sqnorm_workspace_bytes(), W.device)
    lib.check(L.re_rows_sqnorm(_p(W), W.shape[0], W.shape[1], _p(idx), idx.numel(), float(scale), _p(out),
                               int(bool(accumulate)), _p(ws), ws.numel(), _stream()), "re_rows_sqnorm")
    return out


# ------------------------------------------------------------------------------------------------ K10
def adam_step(p, g, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    for t, nme in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nme)
    lib.check(lib.load().re_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), int(step), float(lr), float(beta1),
                                      float(beta2), float(eps), float(weight_decay), _stream()), "re_adam_step")


def adam_step_dev(p, g, m, v, hyper, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """Adam with the step-dependent scalars in device memory: hyper = [lr/(1-b1^t), 1/sqrt(1-b2^t)]  (re_adam_step_dev)."""
    for t, nme in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (hyper, "hyper")):
        _req(t, torch.float32, nme)
    lib.check(lib.load().re_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), float(beta1), float(beta2), float(eps),
                                          float(weight_decay), _stream()), "re_adam_step_dev")


def adam_step_reduce(p, parts, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, gscale=1.0, g_out=None, hyper=None):
    """Adam on gscale * (parts[0] + parts[1] + ... in order); parts [G, n] (row stride >= n), p / m / v [n]  (re_adam_step_reduce: the owner's
    half of a data-parallel step, recboard_amd/dp.py).  step >= 1 with lr, or step = 0 with `hyper` (a captured step's device scalars)."""
    for t, nme in ((p, "p"), (parts, "parts"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nme)
    assert parts.dim() == 2 and parts.stride(1) == 1 and parts.shape[1] == p.numel()
    lib.check(lib.load().re_adam_step_reduce(_p(p), _p(parts), int(parts.shape[0]), int(parts.stride(0)), _p(g_out), _p(m), _p(v), p.numel(), int(step),
                                             float(lr), _p(hyper), float(beta1), float(beta2), float(eps), float(weight_decay), float(gscale), _stream()),
              "re_adam_step_reduce")


def grad_clip_coef(g, max_norm, out=None):
    """-> float32[2] on the device: [min(1, max_norm / (||g|| + 1e-6)), ||g||] over the flat gradient (re_grad_clip_coef: clip_grad_norm_'s
    coefficient, DeepFM/main.py:267)."""
    _req(g, torch.float32, "g")
    out = out if out is not None else torch.empty(2, dtype=torch.float32, device=g.device)
    L = lib.load()
    ws = _ws(L.re_grad_clip_workspace_bytes(), g.device)
    lib.check(L.re_grad_clip_coef(_p(g), g.numel(), float(max_norm), _p(out), _p(ws), ws.numel(), _stream()), "re_grad_clip_coef")
    return out


def adam_step_scaled(p, g, m, v, gscale, step=0, lr=0.0, hyper=None, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """Adam on gscale[0] * g, the scaled gradient written back to g (re_adam_step_scaled).  step >= 1 with lr: host-side bias corrections;
    step = 0 with `hyper` (device float32[2]): a captured step's."""
    for t, nme in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (gscale, "gscale")):
        _req(t, torch.float32, nme)
    lib.check(lib.load().re_adam_step_scaled(_p(p), _p(g), _p(m), _p(v), p.numel(), int(step), float(lr), _p(hyper), float(beta1), float(beta2), float(eps),
                                             float(weight_decay), _p(gscale), _stream()), "re_adam_step_scaled")


def adam_step_clip2(p, g, m, v, n_first, max_norm, step=0, lr=0.0, hyper=None, beta1=0.9, beta2=0.999, eps=1e-8, wd_first=0.0, wd_rest=0.0, out=None):
    """clip_grad_norm_(.., max_norm) + Adam over an arena whose first n_first entries decay with wd_first and the rest with wd_rest
    (re_adam_step_clip2: two launches); g is left clipped.  -> float32[2] on the device: [coefficient, ||g||]."""
    for t, nme in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nme)
    out = out if out is not None else torch.empty(2, dtype=torch.float32, device=g.device)
    L = lib.load()
    ws = _ws(L.re_grad_clip_workspace_bytes(), g.device)
    lib.check(L.re_adam_step_clip2(_p(p), _p(g), _p(m), _p(v), p.numel(), int(n_first), int(step), float(lr), _p(hyper), float(beta1), float(beta2),
                                   float(eps), float(wd_first), float(wd_rest), float(max_norm), _p(out), _p(ws), ws.numel(), _stream()),
              "re_adam_step_clip2")
    return out


def scale_copy(dst, src, alpha):
    _req(dst, torch.float32, "dst"); _req(src, torch.float32, "src")
    lib.check(lib.load().re_scale_copy(_p(dst), _p(src), float(alpha), src.numel(), _stream()), "re_scale_copy")
    return dst


METRIC_NAMES = ("HITRATE", "PRECISION", "RECALL", "NDCG", "MRR")


def rank_metrics(topk_idx, tgt_ptr, tgt_idx, ks):
    """-> (per_user [B, nk, 5], sums [nk, 5]) for METRIC_NAMES at every k in `ks` (re_rank_metrics)."""
    _req(topk_idx, torch.int64, "topk_idx"); _req(tgt_ptr, torch.int64, "tgt_ptr"); _req(tgt_idx, torch.int64, "tgt_idx")
    B, Kmax = topk_idx.shape
    ks = [int(k) for k in ks]
    arr = (ctypes.c_int32 * len(ks))(*ks)
    per_user = torch.empty((B, len(ks), 5), dtype=torch.float32, device=topk_idx.device)
    sums = torch.empty((len(ks), 5), dtype=torch.float32, device=topk_idx.device)
    lib.check(lib.load().re_rank_metrics(_p(topk_idx), B, Kmax, _p(tgt_ptr), _p(tgt_idx), arr, len(ks), _p(per_user), _p(sums),
                                         _stream()), "re_rank_metrics")
    return per_user, sums


def score_pool(Q, E, pool):
    """-> scores f32 [B, P]: <Q[b], E[pool[b, p]]> (re_score_pool: the value the full-catalog score gives the pair, bit for bit)."""
    _req(Q, torch.float32, "Q"); _req(E, torch.float32, "E"); _req(pool, torch.int64, "pool")
    B, D = Q.shape
    if pool.dim() != 2 or pool.shape[0] != B:
        raise ValueError("recengine: `pool` must be [B, P]")
    out = torch.empty(pool.shape, dtype=torch.float32, device=Q.device)
    lib.check(lib.load().re_score_pool(_p(Q), _p(E), _p(pool), B, pool.shape[1], E.shape[0], D, _p(out), _stream()), "re_score_pool")
    return out


def pool_topk(scores, K):
    """-> (vals f32 [B, K] descending, idx int64 [B, K] = positions in the row; ties -> lowest position)  (re_pool_topk)."""
    _req(scores, torch.float32, "scores")
    B, P = scores.shape
    vals = torch.empty((B, K), dtype=torch.float32, device=scores.device)
    idx = torch.empty((B, K), dtype=torch.int64, device=scores.device)
    lib.check(lib.load().re_pool_topk(_p(scores), B, P, K, _p(vals), _p(idx), _stream()), "re_pool_topk")
    return vals, idx


def auc(scores, labels):
    """AUC (Mann-Whitney, ties one half) of `scores` against 0/1 `labels`, exact and sort-free (re_auc).  -> float32[1]."""
    _req(scores, torch.float32, "scores"); _req(labels, torch.float32, "labels")
    out = torch.empty(1, dtype=torch.float32, device=scores.device)
    ws = _ws(256, scores.device)
    lib.check(lib.load().re_auc(_p(scores), _p(labels), scores.numel(), _p(out), _p(ws), ws.numel(), _stream()), "re_auc")
    return out


# ------------------------------------------------------------------------------------------------ K9
def fm_bag_fwd(T, TL, lr_bias, offsets, x, rows_out=None, keys_t=None):
    """-> (E [B, F*D], fm_lr [B])  (re_fm_bag_fwd).  rows_out (int64 [B * F], optional) receives offsets[f] + x[b, f]; keys_t (int32 [F * B],
    optional) x[b, f] field-major (fm_table_grad's keys)."""
    _req(T, torch.float32, "T"); _req(TL, torch.float32, "TL"); _req(lr_bias, torch.float32, "lr_bias")
    _req(offsets, torch.int64, "offsets"); _req(x, torch.int64, "x")
    B, F = x.shape
    D = T.shape[1]
    E = torch.empty((B, F * D), dtype=torch.float32, device=T.device)
    fm_lr = torch.empty((B,), dtype=torch.float32, device=T.device)
    lib.check(lib.load().re_fm_bag_fwd(_p(T), _p(TL), _p(lr_bias), _p(offsets), T.shape[0], _p(x), B, F, D, _p(E), _p(fm_lr),
                                       _p(rows_out), _p(keys_t), _stream()), "re_fm_bag_fwd")
    return E, fm_lr


def fm_bag_bwd(E, dE_mlp, dlogit, F, D):
    """-> (gE [B*F, D], gL [B*F, 1]) contribution rows  (re_fm_bag_bwd)."""
    _req(E, torch.float32, "E"); _req(dlogit, torch.float32, "dlogit")
    if dE_mlp is not None:
        _req(dE_mlp, torch.float32, "dE_mlp")
    B = E.shape[0]
    gE = torch.empty((B * F, D), dtype=torch.float32, device=E.device)
    gL = torch.empty((B * F, 1), dtype=torch.float32, device=E.device)
    lib.check(lib.load().re_fm_bag_bwd(_p(E), _p(dE_mlp), _p(dlogit), B, F, D, _p(gE), _p(gL), _stream()), "re_fm_bag_bwd")
    return gE, gL


def fm_table_grad_ok(B, F, D, max_count):
    """Shapes re_fm_table_grad takes (else: scatter_plan + scatter_apply)."""
    return B <= 8192 and F <= 64 and D <= 15 and max_count < (1 << 19) - 1


def fm_table_slices(counts, B, device, target=40):
    """Row slices for fm_table_grad: int32 [n, 4] = (field, first row, end row, 0), rows relative to the field's first; sized for ~`target` keys a
    slice when a field's B keys are uniform over its rows (a slice of up to 64 keys is one wave's work)."""
    out = []
    for f, c in enumerate(counts):
        rps = max(1, (target * c + B - 1) // max(B, 1))
        for lo in range(0, c, rps):
            out.append((f, lo, min(lo + rps, c), 0))
    return torch.tensor(out, dtype=torch.int32, device=device).reshape(-1, 4)


def fm_table_grad(keys_t, offsets, slices, gE, gL, gT, gTL):
    """gT [R, D], gTL [R] (both ZERO-FILLED by the caller) += the contribution rows of fm_bag_bwd, summed per destination row
    offsets[f] + keys_t[f, b]  (re_fm_table_grad: one launch, a workgroup per row slice: fm_table_slices; bitwise reproducible).
    keys_t: int32 [F, B], the batch's ids field-major (fm_bag_fwd's keys_t)."""
    _req(keys_t, torch.int32, "keys_t"); _req(offsets, torch.int64, "offsets"); _req(gE, torch.float32, "gE"); _req(gL, torch.float32, "gL")
    _req(gT, torch.float32, "gT"); _req(gTL, torch.float32, "gTL"); _req(slices, torch.int32, "slices")
    F = offsets.numel()
    B = keys_t.numel() // F
    R, D = gT.shape
    if gE.numel() != B * F * D or gL.numel() != B * F or gTL.numel() != R or slices.dim() != 2 or slices.shape[1] != 4:
        raise ValueError("fm_table_grad: shapes")
    lib.check(lib.load().re_fm_table_grad(_p(keys_t), B, F, _p(offsets), R, _p(slices), slices.shape[0], _p(gE), _p(gL), D, _p(gT), _p(gTL),
                                          _stream()), "re_fm_table_grad")


def bce_logits(logits, labels):
    """-> (loss [1], dlogit [n], dsum [1]): mean BCE-with-logits and its gradient (re_bce_logits)."""
    _req(logits, torch.float32, "logits"); _req(labels, torch.float32, "labels")
    n = logits.numel()
    dev = logits.device
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dl = torch.empty(n, dtype=torch.float32, device=dev)
    ds = torch.empty(1, dtype=torch.float32, device=dev)
    lib.check(lib.load().re_bce_logits(_p(logits), _p(labels), n, _p(loss), _p(dl), _p(ds), _stream()), "re_bce_logits")
    return loss, dl, ds


# ------------------------------------------------------------------------------------------------ GEMM
def gemm(A, B, transA=False, transB=False, alpha=1.0, beta=0.0, out=None, bias=None, relu=False, defer=None):
    """out = alpha * op(A) @ op(B) + beta * out (+ bias) (+ ReLU) on the fp32 matrix cores (re_gemm_f32).
    A, B: 2-D fp32 with unit inner stride (row stride = leading dimension).
    defer (a list; beta = 0, no bias / relu): a product that is computed split-K leaves its partial products there instead of reducing them --
    `out` is then NOT complete until gemm_reduce_many(defer) has run (one launch for all the deferred products)."""
    for t, nme in ((A, "A"), (B, "B")):
        _req(t, torch.float32, nme, contiguous=False)
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"recengine: `{nme}` must be 2-D with unit inner stride")
    M, K = (A.shape[1], A.shape[0]) if transA else A.shape
    Kb, N = (B.shape[1], B.shape[0]) if transB else B.shape
    if K != Kb:
        raise ValueError(f"recengine: inner dimensions differ ({K} vs {Kb})")
    if out is None:
        if beta != 0.0:
            raise ValueError("recengine: beta != 0 needs `out`")
        out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    _req(out, torch.float32, "out", contiguous=False)
    if bias is not None:
        _req(bias, torch.float32, "bias")
    L = lib.load()
    ws = _ws(L.re_gemm_f32_workspace_bytes(M, N, K), A.device)
    if defer is not None and beta == 0.0 and bias is None and not relu and len(defer) < 8:
        ns = ctypes.c_int32(1)
        lib.check(L.re_gemm_f32_slabs(int(transA), int(transB), M, N, K, float(alpha), _p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0),
                                      _p(ws), ws.numel(), ctypes.byref(ns), _stream()), "re_gemm_f32_slabs")
        if ns.value > 1:
            defer.append((ws, ns.value, M, N, float(alpha), out))
        return out
    lib.check(L.re_gemm_f32(int(transA), int(transB), M, N, K, float(alpha), _p(A), A.stride(0), _p(B), B.stride(0), float(beta),
                            _p(out), out.stride(0), _p(bias), int(relu), _p(ws), ws.numel(), _stream()), "re_gemm_f32")
    return out


def gemm_reduce_many(pending):
    """Finishes the products gemm(..., defer=pending) left as split-K partials: one launch (re_gemm_splitk_reduce_many)."""
    n = len(pending)
    if n == 0:
        return
    slabs = (ctypes.c_void_p * n)(*[_note(p[0]) for p in pending])
    ns = (ctypes.c_int32 * n)(*[p[1] for p in pending])
    Ms = (ctypes.c_int64 * n)(*[p[2] for p in pending])
    Ns = (ctypes.c_int64 * n)(*[p[3] for p in pending])
    al = (ctypes.c_float * n)(*[p[4] for p in pending])
    Cs = (ctypes.c_void_p * n)(*[_note(p[5]) for p in pending])
    ld = (ctypes.c_int64 * n)(*[p[5].stride(0) for p in pending])
    lib.check(lib.load().re_gemm_splitk_reduce_many(n, slabs, ns, Ms, Ns, al, Cs, ld, _stream()), "re_gemm_splitk_reduce_many")
    del pending[:]


def gemm_colstats(A, B, transB=True, bias=None):
    """-> (out, colstats) with out = A @ op(B) + bias and colstats [M / 64, 2, N] = per-64-row (mean, M2) of out's columns, from the GEMM's
    own epilogue (re_gemm_f32_colstats); or None where that form does not apply (M not a multiple of 64, unaligned operands)."""
    for t, nme in ((A, "A"), (B, "B")):
        _req(t, torch.float32, nme, contiguous=False)
        if t.dim() != 2 or t.stride(1) != 1:
            raise ValueError(f"recengine: `{nme}` must be 2-D with unit inner stride")
    M, K = A.shape
    Kb, N = (B.shape[1], B.shape[0]) if transB else B.shape
    if K != Kb:
        raise ValueError(f"recengine: inner dimensions differ ({K} vs {Kb})")
    if M % 64:
        return None
    out = torch.empty((M, N), dtype=torch.float32, device=A.device)
    cs = torch.empty((M // 64, 2, N), dtype=torch.float32, device=A.device)
    if bias is not None:
        _req(bias, torch.float32, "bias")
    rc = lib.load().re_gemm_f32_colstats(0, int(transB), M, N, K, 1.0, _p(A), A.stride(0), _p(B), B.stride(0), _p(out), out.stride(0), _p(bias), _p(cs),
                                         _stream())
    if rc == lib.RE_EUNSUPPORTED:
        return None
    lib.check(rc, "re_gemm_f32_colstats")
    return out, cs


class CEStats:
    """Running row statistics of a chunked cross entropy (re_ce_chunk_*): rowmax, rowsum, target logit per row."""

    def __init__(self, M, device):
        self.rowmax, self.rowsum, self.tgt = (torch.empty(M, dtype=torch.float32, device=device) for _ in range(3))
        self.first = True


def ce_chunk_stats(logits, col0, labels, st):
    """Fold the logits chunk [M, Nc] (catalog columns [col0, col0 + Nc)) into the running statistics `st` (re_ce_chunk_stats)."""
    _req(logits, torch.float32, "logits", contiguous=False); _req(labels, torch.int64, "labels")
    M, Nc = logits.shape
    lib.check(lib.load().re_ce_chunk_stats(_p(logits), M, Nc, logits.stride(0), int(col0), _p(labels), int(st.first), _p(st.rowmax),
                                           _p(st.rowsum), _p(st.tgt), _stream()), "re_ce_chunk_stats")
    st.first = False


def ce_chunk_loss(st, labels, N):
    """mean over the rows of logsumexp - target logit, after the last chunk (re_ce_chunk_loss) -> loss[1]."""
    M = labels.numel()
    row_loss = torch.empty(M, dtype=torch.float32, device=labels.device)
    loss = torch.empty(1, dtype=torch.float32, device=labels.device)
    lib.check(lib.load().re_ce_chunk_loss(_p(st.rowmax), _p(st.rowsum), _p(st.tgt), _p(labels), M, int(N), _p(row_loss), _p(loss), _stream()),
              "re_ce_chunk_loss")
    return loss


def ce_chunk_grad_(logits, col0, labels, st):
    """In place: a recomputed logits chunk -> d(mean CE)/d(that chunk's logits) (re_ce_chunk_grad)."""
    _req(logits, torch.float32, "logits", contiguous=False)
    M, Nc = logits.shape
    lib.check(lib.load().re_ce_chunk_grad(_p(logits), M, Nc, logits.stride(0), int(col0), _p(labels), _p(st.rowmax), _p(st.rowsum), _stream()),
              "re_ce_chunk_grad")
    return logits


def ce_rows_(logits, labels):
    """In place: logits [M, N] -> d(mean CE)/d(logits); returns loss [1]  (re_ce_rows)."""
    _req(logits, torch.float32, "logits"); _req(labels, torch.int64, "labels")
    M, N = logits.shape
    row_loss = torch.empty(M, dtype=torch.float32, device=logits.device)
    loss = torch.empty(1, dtype=torch.float32, device=logits.device)
    lib.check(lib.load().re_ce_rows(_p(logits), M, N, logits.stride(0), _p(labels), _p(row_loss), _p(loss), _stream()), "re_ce_rows")
    return loss


# ------------------------------------------------------------------------------------------------ MLP pieces
def step_state(state, seed, step, lr, beta1=0.9, beta2=0.999):
    """The per-step device words of a captured step (re_step_state): state int32[4] = {seed, 0, Adam step size, 1 / sqrt(bias correction 2)}."""
    _req(state, torch.int32, "state")
    lib.check(lib.load().re_step_state(_p(state), int(seed) & 0xFFFFFFFF, int(step), float(lr), float(beta1), float(beta2), _stream()), "re_step_state")
    return state


def stage_inputs(state, seed, step, lr, beta1, beta2, pairs=(), zeros=()):
    """step_state + the batch into a captured step's static buffers + zero fills, ONE launch (re_step_stage_inputs).  pairs: (static, input) with
    the same dtype (a byte copy) or fp32 <- int64 (a cast); zeros: tensors to clear.  Anything else (another dtype pair, a non-contiguous
    tensor, more than 8 segments) goes the ordinary way: copy_ / zero_."""
    import ctypes
    segs = []
    for s, x in pairs:
        if s.is_contiguous() and x.is_contiguous() and x.is_cuda and s.numel() == x.numel() and x.dtype == s.dtype:
            segs.append((s, x, 0))
        elif s.is_contiguous() and x.is_contiguous() and x.is_cuda and s.numel() == x.numel() and s.dtype == torch.float32 and x.dtype == torch.int64:
            segs.append((s, x, 1))
        else:
            s.copy_(x, non_blocking=True)
    for z in zeros:
        if z.is_contiguous():
            segs.append((z, None, 2))
        else:
            z.zero_()
    while len(segs) > 8:
        s, x, kind = segs.pop()
        if kind == 2:
            s.zero_()
        else:
            s.copy_(x, non_blocking=True)
    n = len(segs)
    for s, x, _ in segs:
        _note(s)
        if x is not None:
            _note(x)
    dst = (ctypes.c_void_p * max(n, 1))(*[s.data_ptr() for s, _, _ in segs])
    src = (ctypes.c_void_p * max(n, 1))(*[(x.data_ptr() if x is not None else 0) for _, x, _ in segs])
    nb = (ctypes.c_int64 * max(n, 1))(*[s.numel() * s.element_size() for s, _, _ in segs])
    kd = (ctypes.c_int32 * max(n, 1))(*[k for _, _, k in segs])
    if state is not None:
        _req(state, torch.int32, "state")
    lib.check(lib.load().re_step_stage_inputs(_p(state), int(seed) & 0xFFFFFFFF, int(step), float(lr), float(beta1), float(beta2), n, dst, src, nb, kd,
                                              _stream()), "re_step_stage_inputs")


def bn_relu_drop_fwd(z, gamma, beta, run_mean, run_var, training, drop_p=0.0, seed=0, stream_id=100, eps=1e-5, momentum=0.1, seed_dev=None,
                     colstats=None):
    """-> (a, stats): a = dropout(relu(bn(z)))  (re_bn_relu_drop_fwd); gamma None = no BatchNorm.  colstats (training, BatchNorm): the
    per-chunk (mean, M2) partials of z's columns from `gemm_colstats` -- the statistics pass over z is skipped (re_bn_relu_drop_fwd_pre)."""
    _req(z, torch.float32, "z")
    M, N = z.shape
    a = torch.empty_like(z)
    stats = torch.empty(2 * N, dtype=torch.float32, device=z.device) if gamma is not None else None
    L = lib.load()
    if colstats is not None and training and gamma is not None:
        lib.check(L.re_bn_relu_drop_fwd_pre(_p(z), M, N, _p(gamma), _p(beta), _p(run_mean), _p(run_var), float(eps), float(momentum), float(drop_p),
                                            int(seed) & 0xFFFFFFFF, _p(seed_dev), int(stream_id), _p(stats), _p(a), _p(colstats), int(colstats.shape[0]),
                                            _stream()), "re_bn_relu_drop_fwd_pre")
        return a, stats
    ws = _ws(L.re_mlp_workspace_bytes(N), z.device)
    lib.check(L.re_bn_relu_drop_fwd(_p(z), M, N, _p(gamma), _p(beta), _p(run_mean), _p(run_var), int(bool(training)),
                                    float(eps), float(momentum), float(drop_p), int(seed) & 0xFFFFFFFF, _p(seed_dev), int(stream_id),
                                    _p(stats), _p(a), _p(ws), ws.numel(), _stream()), "re_bn_relu_drop_fwd")
    return a, stats


def bn_relu_drop_bwd(da, a, z, gamma, stats, drop_p, dgamma=None, dbeta=None):
    """-> (dz, dgamma, dbeta)  (re_bn_relu_drop_bwd)."""
    _req(da, torch.float32, "da")
    M, N = z.shape
    dz = torch.empty_like(z)
    if dbeta is None:
        dbeta = torch.empty(N, dtype=torch.float32, device=z.device)
    if gamma is not None and dgamma is None:
        dgamma = torch.empty(N, dtype=torch.float32, device=z.device)
    L = lib.load()
    ws = _ws(L.re_mlp_workspace_bytes(N), z.device)
    lib.check(L.re_bn_relu_drop_bwd(_p(da), _p(a), _p(z), M, N, _p(gamma), _p(stats), float(drop_p), _p(dz), _p(dgamma),
                                    _p(dbeta), _p(ws), ws.numel(), _stream()), "re_bn_relu_drop_bwd")
    return dz, dgamma, dbeta


def mlp_head_fwd(h, w, b, fm_lr=None, labels=None, dsum=None, dsum2=None, defer_final=False):
    """DeepFM's Linear(., 1) + logit sum (+ criterion) (re_mlp_head_fwd): -> logits [M], and with labels (loss [1], dlogit [M], dsum [1]);
    dsum2: a second place for sum dlogit (DeepFM's LR bias has the same gradient as the last layer's).
    defer_final: loss / dsum / dsum2 are NOT written by this call -- a fifth return value, the workspace with the per-workgroup partials, goes
    to mlp_head_bwd_gated(final=...), whose launch adds them (one dispatch less)."""
    _req(h, torch.float32, "h"); _req(w, torch.float32, "w"); _req(b, torch.float32, "b")
    M, K = h.shape
    dev = h.device
    logits = torch.empty(M, dtype=torch.float32, device=dev)
    L = lib.load()
    ws = _ws(L.re_mlp_head_workspace_bytes(M, K), dev)
    if labels is None:
        lib.check(L.re_mlp_head_fwd(_p(h), M, K, _p(w), _p(b), _p(fm_lr), None, _p(logits), None, None, None, None, _p(ws), ws.numel(), _stream()), "re_mlp_head_fwd")
        return logits
    _req(labels, torch.float32, "labels")
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    dl = torch.empty(M, dtype=torch.float32, device=dev)
    dsum = dsum if dsum is not None else torch.empty(1, dtype=torch.float32, device=dev)
    lib.check(L.re_mlp_head_fwd(_p(h), M, K, _p(w), _p(b), _p(fm_lr), _p(labels), _p(logits), None if defer_final else _p(loss), _p(dl), _p(dsum),
                                _p(dsum2), _p(ws), ws.numel(), _stream()), "re_mlp_head_fwd")
    if defer_final:
        return logits, loss, dl, dsum, (ws, loss, dsum, dsum2)
    return logits, loss, dl, dsum


def mlp_head_bwd(dlogit, h, w, dW):
    """-> da [M, K] = dlogit (x) w; dW [K] (+)= nothing: written = sum_m dlogit[m] h[m, :]  (re_mlp_head_bwd)."""
    _req(dlogit, torch.float32, "dlogit"); _req(h, torch.float32, "h"); _req(w, torch.float32, "w"); _req(dW, torch.float32, "dW")
    M, K = h.shape
    da = torch.empty_like(h)
    L = lib.load()
    ws = _ws(L.re_mlp_head_workspace_bytes(M, K), h.device)
    lib.check(L.re_mlp_head_bwd(_p(dlogit), _p(h), _p(w), M, K, _p(da), _p(dW), _p(ws), ws.numel(), _stream()), "re_mlp_head_bwd")
    return da


def mlp_head_bwd_gated(dlogit, h, w, z, stats, drop_p, final=None):
    """The last Linear(., 1)'s backward with the gate of the block underneath (re_mlp_head_bwd_gated): -> (g [M, K], part [chunks, 3, K]).
    final: mlp_head_fwd(defer_final=True)'s fifth return value -- the criterion's loss / sum dlogit are finished by this launch."""
    _req(dlogit, torch.float32, "dlogit"); _req(h, torch.float32, "h"); _req(w, torch.float32, "w"); _req(z, torch.float32, "z"); _req(stats, torch.float32, "stats")
    M, K = h.shape
    g = torch.empty_like(h)
    L = lib.load()
    nbytes = L.re_mlp_head_workspace_bytes(M, K)
    part = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=h.device)
    ch = ctypes.c_int(0)
    fw, fl, fd, fd2 = final if final is not None else (None, None, None, None)
    lib.check(L.re_mlp_head_bwd_gated(_p(dlogit), _p(h), _p(w), M, K, _p(z), _p(stats), float(drop_p), _p(g), _p(part), part.numel() * 4, ctypes.byref(ch),
                                      _p(fw), _p(fl), _p(fd), _p(fd2), _stream()), "re_mlp_head_bwd_gated")
    return g, part[:ch.value * 3 * K].view(ch.value, 3, K)


def gemm_gated(A, B, act, z, stats, drop_scale, transB=False):
    """g = act > 0 ? drop_scale A op(B) : 0 and its per-64-row column sums (re_gemm_f32_gated) -> (g [M, N], part [M / 64, 2, N]), or None where
    the form does not apply (M not a multiple of 64, unaligned operands)."""
    _req(A, torch.float32, "A"); _req(B, torch.float32, "B"); _req(act, torch.float32, "act"); _req(z, torch.float32, "z"); _req(stats, torch.float32, "stats")
    M, K = A.shape
    N = B.shape[0] if transB else B.shape[1]
    if M % 64 or act.shape != (M, N) or z.shape != (M, N):
        return None
    g = torch.empty((M, N), dtype=torch.float32, device=A.device)
    part = torch.empty((M // 64, 2, N), dtype=torch.float32, device=A.device)
    rc = lib.load().re_gemm_f32_gated(0, int(transB), M, N, K, 1.0, _p(A), A.stride(0), _p(B), B.stride(0), _p(g), N, _p(act), _p(z), _p(stats),
                                      float(drop_scale), _p(part), _stream())
    if rc == lib.RE_EUNSUPPORTED:
        return None
    lib.check(rc, "re_gemm_f32_gated")
    return g, part


def bn_bwd_apply(g, z, gamma, stats, part, dgamma, dbeta, extra_out=None):
    """g -> dz in place, dgamma / dbeta (/ extra_out) from the per-chunk partials (re_bn_bwd_apply)."""
    _req(g, torch.float32, "g"); _req(z, torch.float32, "z"); _req(part, torch.float32, "part")
    M, N = z.shape
    lib.check(lib.load().re_bn_bwd_apply(_p(g), _p(z), M, N, _p(gamma), _p(stats), _p(part), int(part.shape[0]), int(part.shape[1]), _p(dgamma), _p(dbeta),
                                         _p(extra_out), _stream()), "re_bn_bwd_apply")
    return g


def colsum(x, out=None):
    _req(x, torch.float32, "x")
    M, N = x.shape
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=x.device)
    L = lib.load()
    ws = _ws(L.re_mlp_workspace_bytes(N), x.device)
    lib.check(L.re_colsum(_p(x), M, N, _p(out), _p(ws), ws.numel(), _stream()), "re_colsum")
    return out
