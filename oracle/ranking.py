"""Oracle: full-catalog scoring, seen-mask, top-K and ranking metrics (TEST INFRASTRUCTURE).

Contract = `freerec.launcher.Coach.evaluate` (external, PARITY UNPINNED), mirrored at UniSRec/main.py:400-447:
    scores = model(data, ranking="full")            # [B, N] = U . E^T          (SASRec/main.py:223-228)
    scores[seen] = -1e23                             # if not cfg.retain_seen
    targets = Item.to_csr(data[IUnseen]).to_dense()  # [B, N] 0/1
    monitor(scores, targets, pool=[HITRATE, PRECISION, RECALL, NDCG, MRR]) for every NAME@K in cfg.monitors
    (cfg.monitors on the benchmark: HitRate@{1,5,10,20,50}, NDCG@{5,10,20,50}; SASRec/configs/Amazon2014Beauty_550_LOU.yaml:21)

Metric definitions are freerec's (not in /root/reference) -> restated in their textbook forms; the known-answer
relations the published rows satisfy (one target per user => HR@1 == NDCG@1, NDCG@K <= HR@K,
NDCG@K >= HR@K / log2(K+1)) are tested in tests/test_oracle_golden.py.

The score arithmetic is done by the C oracle (oracle/c/recoracle.c): k-ordered fmaf chain, ties -> lowest index.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("RECORACLE_LIB") or os.path.join(_HERE, "_build", "librecoracle.so")   # (RECORACLE_LIB: the sanitizer build, tests only)
        if not os.path.exists(path):
            import subprocess
            subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = ctypes.CDLL(path)
        fp, ip, i64 = ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64
        L.ro_score_dense.argtypes = [fp, fp, i64, i64, i64, fp]
        L.ro_score_topk.argtypes = [fp, fp, i64, i64, i64, ip, ip, i64, fp, ip]
        L.ro_gather_rows.argtypes = [fp, ip, i64, i64, fp]
        L.ro_scatter_add_rows.argtypes = [fp, ip, i64, i64, i64, i64, fp]
        for f in (L.ro_score_dense, L.ro_score_topk, L.ro_gather_rows, L.ro_scatter_add_rows):
            f.restype = None
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def score_dense(Q: np.ndarray, E: np.ndarray) -> np.ndarray:
    Q = np.ascontiguousarray(Q, np.float32)
    E = np.ascontiguousarray(E, np.float32)
    out = np.empty((Q.shape[0], E.shape[0]), np.float32)
    lib().ro_score_dense(_p(Q), _p(E), Q.shape[0], E.shape[0], Q.shape[1], _p(out))
    return out


def score_topk(Q, E, seen_ptr, seen_idx, K):
    """-> (vals f32 [B,K] sorted descending, idx int64 [B,K]); masked items carry -1e23 (float32)."""
    Q = np.ascontiguousarray(Q, np.float32)
    E = np.ascontiguousarray(E, np.float32)
    B = Q.shape[0]
    vals = np.empty((B, K), np.float32)
    idx = np.empty((B, K), np.int64)
    if seen_ptr is None:
        sp = si = None
    else:
        sp = np.ascontiguousarray(seen_ptr, np.int64)
        si = np.ascontiguousarray(seen_idx, np.int64)
    lib().ro_score_topk(_p(Q), _p(E), B, E.shape[0], Q.shape[1],
                        _p(sp) if sp is not None else None, _p(si) if si is not None else None,
                        K, _p(vals), _p(idx))
    return vals, idx


def score_pool(Q: np.ndarray, E: np.ndarray, pool: np.ndarray) -> np.ndarray:
    """recommend_from_pool (SASRec/main.py:230-236 einsum("BD,BKD->BK"); MF-BPR/main.py:106-109; LightGCN/main.py:122-125): scores [B, P] of
    every row's candidate pool -- the entries of `score_dense` (the C fmaf chain) at the pool's columns."""
    full = score_dense(np.ascontiguousarray(Q, np.float32), np.ascontiguousarray(E, np.float32))
    return np.take_along_axis(full, np.asarray(pool, np.int64), axis=1)


def pool_topk(scores: np.ndarray, K: int):
    """Top-K of every row of a [B, P] pool score matrix, ties to the lowest position (the full ranking's rule; evaluate contract
    UniSRec/main.py:415-421 puts the target at position 0); slots beyond P: (-inf, -1)."""
    B, P = scores.shape
    order = np.argsort(-scores.astype(np.float64), axis=1, kind="stable")
    vals = np.full((B, K), -np.inf, np.float32)
    idx = np.full((B, K), -1, np.int64)
    k = min(K, P)
    idx[:, :k] = order[:, :k]
    vals[:, :k] = np.take_along_axis(scores, order[:, :k], axis=1)
    return vals, idx


def gather_rows_c(W, idx):
    W = np.ascontiguousarray(W, np.float32)
    ix = np.ascontiguousarray(idx, np.int64).reshape(-1)
    out = np.empty((ix.shape[0], W.shape[1]), np.float32)
    lib().ro_gather_rows(_p(W), _p(ix), ix.shape[0], W.shape[1], _p(out))
    return out.reshape(tuple(np.shape(idx)) + (W.shape[1],))


def scatter_add_rows_c(g, idx, R, padding_idx=-1):
    D = g.shape[-1]
    g2 = np.ascontiguousarray(g, np.float32).reshape(-1, D)
    ix = np.ascontiguousarray(idx, np.int64).reshape(-1)
    out = np.empty((R, D), np.float32)
    lib().ro_scatter_add_rows(_p(g2), _p(ix), ix.shape[0], D, R, padding_idx, _p(out))
    return out


# ---------------------------------------------------------------- metrics from a sorted top-K list
def metrics_from_topk(topk_idx: np.ndarray, tgt_ptr: np.ndarray, tgt_idx: np.ndarray, ks=(1, 5, 10, 20, 50)):
    """Per-user metric arrays {NAME@K: float64[B]} from top-K item ids (sorted by rank) and ragged targets."""
    B, Kmax = topk_idx.shape
    out = {}
    hits = np.zeros((B, Kmax), bool)
    ntgt = np.zeros(B, np.int64)
    for b in range(B):
        t = tgt_idx[tgt_ptr[b]:tgt_ptr[b + 1]]
        ntgt[b] = len(t)
        hits[b] = np.isin(topk_idx[b], t)
    disc = 1.0 / np.log2(np.arange(Kmax) + 2.0)
    for k in ks:
        if k > Kmax:
            continue
        h = hits[:, :k]
        nh = h.sum(1)
        out[f"HITRATE@{k}"] = (nh > 0).astype(np.float64)
        out[f"PRECISION@{k}"] = nh / float(k)
        out[f"RECALL@{k}"] = nh / np.maximum(ntgt, 1)
        dcg = (h * disc[:k]).sum(1)
        idcg = np.array([disc[:min(k, max(int(n), 1))].sum() for n in ntgt])
        out[f"NDCG@{k}"] = dcg / idcg
        first = np.where(h.any(1), h.argmax(1), -1)
        out[f"MRR@{k}"] = np.where(first >= 0, 1.0 / (np.maximum(first, 0) + 1.0), 0.0)
    return out


def metrics_dense(scores: np.ndarray, targets: np.ndarray, ks=(1, 5, 10, 20, 50)):
    """Same metrics from dense [B,N] scores / 0-1 targets (the shape Coach.evaluate hands to monitor())."""
    B, N = scores.shape
    Kmax = min(max(ks), N)
    order = np.lexsort((np.arange(N)[None, :].repeat(B, 0), -scores.astype(np.float64)), axis=1)[:, :Kmax]
    ptr = np.zeros(B + 1, np.int64)
    rows, cols = np.nonzero(targets)
    np.cumsum(np.bincount(rows, minlength=B), out=ptr[1:])
    return metrics_from_topk(order, ptr, cols.astype(np.int64), ks)
