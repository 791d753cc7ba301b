"""Adjacency construction for the GCN family: host-side mirror of `dataset.train().to_normalized_adj(normalization="sym")`
(LightGCN/main.py:47-49; freerec.graph is external -> parity unpinned, NGCF/main.py:76-87 shows self loops are NOT default).
Runs once at model construction, on the host."""
import numpy as np


def to_normalized_adj(num_users, num_items, edges_u, edges_i, normalization="sym"):
    """Bipartite user-item graph -> CSR (crow int64[n+1], col int64[nnz], val f32[nnz]) of D^-1/2 A D^-1/2 ("sym")
    or D^-1 A ("left"); rows ascending, columns ascending inside a row; duplicate edges are merged."""
    n = num_users + num_items
    eu, ei = np.asarray(edges_u, np.int64), np.asarray(edges_i, np.int64)
    key = np.unique(eu * num_items + ei)
    eu, ei = key // num_items, key % num_items
    rows = np.concatenate([eu, ei + num_users])
    cols = np.concatenate([ei + num_users, eu])
    deg = np.bincount(rows, minlength=n).astype(np.float64)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    if normalization == "sym":
        dinv = np.zeros_like(deg)
        dinv[deg > 0] = deg[deg > 0] ** -0.5
        val = dinv[rows] * dinv[cols]
    elif normalization == "left":
        val = np.where(deg > 0, 1.0 / np.maximum(deg, 1), 0.0)[rows]
    else:
        raise NotImplementedError(f"unknown normalization {normalization!r}")
    crow = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(rows, minlength=n), out=crow[1:])
    return crow, cols, val.astype(np.float32)
