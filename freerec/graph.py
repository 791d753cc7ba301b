"""`freerec.graph`: adjacency helpers (NGCF/main.py:76-87 `add_self_loops` + `to_normalized`; LightGCN's `to_normalized_adj`)."""
import torch


def to_undirected(edge_index, edge_weight=None, num_nodes=None):
    """Both directions of every edge (SGL/main.py:103-105, NGCF/main.py:76): edge_index alone, or (edge_index, edge_weight) when weights
    are given (the torch_geometric convention the call sites follow)."""
    ei = torch.cat([edge_index, edge_index.flip(0)], dim=1)
    if edge_weight is None:
        return ei
    return ei, torch.cat([edge_weight, edge_weight])


def add_self_loops(edge_index, num_nodes=None):
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    loops = torch.arange(n, dtype=edge_index.dtype, device=edge_index.device)
    return torch.cat([edge_index, torch.stack([loops, loops])], dim=1), None


def to_normalized(edge_index, edge_weight=None, normalization="sym", num_nodes=None):
    """D^-1/2 A D^-1/2 ("sym"), D^-1 A ("left") or A D^-1 ("right") edge weights of a graph given as edge_index [2, E]."""
    row, col = edge_index
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    w = torch.ones(row.numel(), dtype=torch.float32, device=row.device) if edge_weight is None else edge_weight
    deg = torch.zeros(n, dtype=w.dtype, device=w.device).index_add_(0, row, w)
    if normalization == "sym":
        d = deg.clamp_min(1e-12).pow(-0.5)
        d[deg == 0] = 0
        w = d[row] * w * d[col]
    elif normalization == "left":
        d = deg.clamp_min(1e-12).reciprocal()
        d[deg == 0] = 0
        w = d[row] * w
    elif normalization == "right":
        degc = torch.zeros(n, dtype=w.dtype, device=w.device).index_add_(0, col, w)
        d = degc.clamp_min(1e-12).reciprocal()
        d[degc == 0] = 0
        w = w * d[col]
    else:
        raise NotImplementedError(f"normalization {normalization!r}")
    return edge_index, w


def to_adjacency(edge_index, edge_weight=None, num_nodes=None):
    """-> torch sparse CSR [n, n] (rows sorted, duplicate edges summed)."""
    n = int(edge_index.max()) + 1 if num_nodes is None else num_nodes
    w = torch.ones(edge_index.shape[1], dtype=torch.float32, device=edge_index.device) if edge_weight is None else edge_weight
    return torch.sparse_coo_tensor(edge_index, w, (n, n)).coalesce().to_sparse_csr()
