"""Oracle: LightGCN encode / fit (TEST INFRASTRUCTURE).  torch-CPU fp32.

  encode  LightGCN/main.py:77-86   X0 = cat(U, I); X_{l+1} = Adj @ X_l; out = sum_l X_l / (L+1), accumulated
                                   incrementally in the reference's order (avg += X_l / (L+1))
  fit     LightGCN/main.py:88-108  BPR on propagated rows + emb_loss = regularize([U0[u], I0[i+], I0[i-]], l2) / B
  step    LightGCN/main.py:160     loss = rec_loss + weight_decay * emb_loss (optimizer built WITHOUT weight decay, :139-145)

Adj = `dataset.train().to_normalized_adj("sym")` (LightGCN/main.py:47-49): symmetric-normalised bipartite adjacency
D^-1/2 A D^-1/2 as a CSR tensor -- freerec-side (PARITY UNPINNED; no self loops, cf. NGCF/main.py:76-87).
"""
import numpy as np
import torch

from . import criterions


def sym_normalized_adj(U: int, N: int, edges_u: np.ndarray, edges_i: np.ndarray):
    """-> (crow int64[U+N+1], col int64[nnz], val f32[nnz]), rows sorted, columns ascending within a row."""
    n = U + N
    rows = np.concatenate([edges_u, edges_i + U]).astype(np.int64)
    cols = np.concatenate([edges_i + U, edges_u]).astype(np.int64)
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    dinv = np.where(deg > 0, deg ** -0.5, 0.0)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    val = (dinv[rows] * dinv[cols]).astype(np.float32)
    crow = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=crow[1:])
    return crow, cols, val


def spmm_csr(crow, col, val, X):
    """Y = A @ X for CSR A; per output row, products accumulated in column order (fp32)."""
    A = torch.sparse_csr_tensor(torch.as_tensor(crow), torch.as_tensor(col), torch.as_tensor(val),
                                size=(len(crow) - 1, X.shape[0]))
    return A @ X


def encode(U, I, crow, col, val, num_layers=3):
    allE = torch.cat((U, I), 0)
    avg = allE / (num_layers + 1)
    for _ in range(num_layers):
        allE = spmm_csr(crow, col, val, allE)
        avg = avg + allE / (num_layers + 1)
    return torch.split(avg, (U.shape[0], I.shape[0]))


def fit(U, I, crow, col, val, users, pos, neg, num_layers=3):
    ue, ie = encode(U, I, crow, col, val, num_layers)
    u, ip, ineg = ue[users], ie[pos], ie[neg]
    rec = criterions.bpr_loss((u * ip).sum(-1), (u * ineg).sum(-1))
    emb = criterions.regularize_l2([U[users], I[pos], I[neg]]) / len(users)
    return rec, emb
