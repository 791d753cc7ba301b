"""Oracle: embedding row gather and its scatter-add gradient (TEST INFRASTRUCTURE).

Reference ops restated:
  * `nn.Embedding.__call__` / `W[idx]`       SASRec/main.py:183,200-204; MF-BPR/main.py:84-86;
                                             LightGCN/main.py:91-93,101-103; DeepFM/main.py:59-61,204-206
  * backward = dense `[R, D]` gradient, rows of `padding_idx` zero (aten embedding_dense_backward)
  * SASRec embedding front end               SASRec/main.py:181-187
        x = E[seq] * sqrt(D) + P[0:S]; x[seq == 0] = 0
"""
import numpy as np


def gather_rows(W: np.ndarray, idx: np.ndarray) -> np.ndarray:
    """out[..., :] = W[idx[...], :]  (exact copy, any dtype)."""
    idx = np.asarray(idx)
    return W[idx.reshape(-1)].reshape(idx.shape + (W.shape[1],)).copy()


def scatter_add_rows(grad_out: np.ndarray, idx: np.ndarray, R: int, padding_idx: int = -1) -> np.ndarray:
    """Dense [R, D] gradient of gather_rows: contributions summed IN POSITION ORDER per destination
    row (the deterministic order the HIP kernel reproduces bit-for-bit), rows == padding_idx stay 0."""
    D = grad_out.shape[-1]
    g = grad_out.reshape(-1, D)
    ix = np.asarray(idx).reshape(-1)
    out = np.zeros((R, D), np.float32)
    for i in range(ix.shape[0]):  # sequential fp32 adds, position order
        r = int(ix[i])
        if r == padding_idx:
            continue
        out[r] = out[r] + g[i]
    return out


def scatter_add_rows_fast(grad_out: np.ndarray, idx: np.ndarray, R: int, padding_idx: int = -1) -> np.ndarray:
    """Same result as scatter_add_rows (np.add.at applies updates sequentially in index order)."""
    D = grad_out.shape[-1]
    g = grad_out.reshape(-1, D).astype(np.float32)
    ix = np.asarray(idx).reshape(-1)
    keep = ix != padding_idx
    out = np.zeros((R, D), np.float32)
    np.add.at(out, ix[keep], g[keep])
    return out


def sasrec_embed(E: np.ndarray, P: np.ndarray, seq: np.ndarray) -> np.ndarray:
    """SASRec/main.py:181-187 with dropout off: (E[seq] * sqrt(D) + P[s]) masked to 0 where seq == 0."""
    B, S = seq.shape
    D = E.shape[1]
    x = E[seq].astype(np.float32) * np.float32(D ** 0.5)
    x = x + P[None, :S, :]
    x[seq == 0] = 0.0
    return x
