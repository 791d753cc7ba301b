"""Sibling baselines on the engine's torch custom ops (recboard_amd.nn / torch.ops.recengine): the reference's model code shape --
an nn.Module with `encode` / `fit` / `recommend_from_*` -- with the hot-path operators swapped for the HIP kernels and autograd
left to do the bookkeeping.  This is the drop-in path a RecBoard model file takes; SASRec / LightGCN / MF-BPR /
DeepFM additionally have hand-fused engines (sasrec.py, gen.py, deepfm.py).

  DCN     (DCN/main.py:34-190)     multi-field embedding lookup = recengine::gather_rows on ONE concatenated table (+ scatter-add
                                   gradient), every Linear (MLP blocks, cross weights, fc) = recengine::gemm forward and backward
  SimGCL  (SimGCL/main.py:34-160)  K propagations = recengine::spmm_csr (symmetric adjacency: the backward is the same kernel),
                                   BPR over the propagated tables = recengine::bpr_triplet (gathers + dots + criterion fused),
                                   the two B x B InfoNCE logit matrices = recengine::score_dense, full ranking = score_dense
  NGCF    (NGCF/main.py:29-160)    K convolutions on the LEFT-normalised adjacency with self loops (not symmetric: nn.spmm runs the backward on
                                   the transposed CSR), both Linear maps per layer = recengine::gemm, BPR over the concatenated layer outputs =
                                   recengine::bpr_triplet, full ranking = recengine::gemm
  GRU4Rec (GRU4Rec/main.py:30-190) item lookup = recengine::gather_rows (padding row without gradient; scatter-add gradient), the dense
                                   projection = recengine::gemm, the pair criteria = recengine::bpr_triplet (BPR: gathers + dots +
                                   softplus fused) or gathers + row dots (BCE), CE and full ranking over the catalog = score_dense
  SGL     (SGL/main.py:30-215)     LightGCN-style propagation on the full graph and on two sampled subgraphs (node / edge dropout, random walk):
                                   a subgraph keeps the full graph's CSR pattern and only re-weights its entries (dropped edge = 0, degrees
                                   from the kept weights), so every propagation is recengine::spmm_csr on the cached plan; BPR =
                                   recengine::bpr_triplet, the two B x B InfoNCE logit matrices = recengine::score_dense
  JGCF    (JGCF/main.py:39-160, modules.py:8-83) the Jacobi-polynomial recurrence = one recengine::spmm_csr per order (symmetric adjacency),
                                   BPR over the [low-pass | mid-pass] tables = recengine::bpr_triplet, full ranking = recengine::gemm
  GCN     (GCN/main.py:27-135)     per layer recengine::spmm_csr then the layer's Linear = recengine::gemm (ReLU between layers), BPR =
                                   recengine::bpr_triplet, full ranking = recengine::gemm
  STAMP   (STAMP/main.py:28-160)   item lookup = recengine::gather_rows, the trilinear attention's w0..w3 and both MLP branches = recengine::gemm,
                                   criteria / ranking as GRU4Rec
  NARM    (NARM/main.py:30-180)    item lookup = recengine::gather_rows, a_1 / a_2 / v_t / b = recengine::gemm, the state at the last real position
                                   = recengine::gather_rows, BCE on gathered rows; the GRU is torch.nn.GRU (MIOpen)
  FMLP-Rec (FMLP-Rec/main.py:38-180, modules.py:27-120) item lookup = recengine::gather_rows, dense_1 / dense_2 of every block = recengine::gemm,
                                   criteria / ranking as GRU4Rec; the frequency-domain filter is torch.fft (rocFFT)
  BSARec  (BSARec/main.py:38-200, modules.py:28-200) item lookup = recengine::gather_rows; query / key / value / dense of the attention branch and
                                   dense_1 / dense_2 of the feed-forward = recengine::gemm; criteria / ranking as GRU4Rec; the low-pass branch is
                                   torch.fft (rocFFT), the S x S attention of a 50-item sequence stays with aten
  BERT4Rec (BERT4Rec/main.py:33-195) item lookup = recengine::gather_rows (padding row without gradient), the encoder states at the MASKED
                                   positions = recengine::gather_rows, the projection to the N + 2 logits (`fc`) = recengine::gemm on those
                                   rows only (the reference projects all B*S rows and then selects), full ranking = recengine::gemm on the
                                   last position.  The bidirectional encoder itself is torch.nn.TransformerEncoder, as in the reference.
Elementwise glue (BatchNorm / ReLU / dropout, the cross layer's x0 * s + b, L2 normalisation, log-softmax of the B x B logits) and
the GRU recurrence itself (torch.nn.GRU: MIOpen) stay with aten: no table, no catalog and no catalog-sized contraction is touched there.
"""
import torch
import torch.nn.functional as F

from . import nn as rnn


# ------------------------------------------------------------------------------------------------ evaluation contract
class _BufferRanking:
    """`recommend_topk` for the graph models: Coach.evaluate's scores -> seen mask -> top-K (UniSRec/main.py:408-414) as ONE fused
    launch on the ranking buffers (`reset_ranking_buffers` first, as the reference's Coach does)."""

    def recommend_topk(self, users, seen_ptr, seen_idx, K=50):
        ue, ie = self.ranking_buffer
        with torch.no_grad():
            return torch.ops.recengine.score_topk(rnn.gather_rows(ue, users.reshape(-1)), ie, seen_ptr, seen_idx, K)


class _EncodeRanking:
    """`recommend_topk` for the sequence models whose `encode` returns (user states [B, D], item table [N, D])."""

    def recommend_topk(self, seqs, seen_ptr, seen_idx, K=50):
        with torch.no_grad():
            user, items = self.encode(seqs)
            return torch.ops.recengine.score_topk(user.contiguous(), items.contiguous(), seen_ptr, seen_idx, K)


# ------------------------------------------------------------------------------------------------ DCN
class CrossInteraction(torch.nn.Module):
    """x_{i+1} = (x_i w) * x_0 + b   (DCN/main.py:34-46)."""

    def __init__(self, input_dim, device=None):
        super().__init__()
        self.weight = rnn.Linear(input_dim, 1, bias=False, device=device)
        self.bias = torch.nn.Parameter(torch.zeros(input_dim, device=device))

    def forward(self, X_0, X_i):
        return self.weight(X_i) * X_0 + self.bias


class MLPBlock(torch.nn.Module):
    """Linear -> BatchNorm1d -> ReLU -> Dropout   (DCN/main.py:48-69; DeepFM/main.py:103-124)."""

    def __init__(self, input_dim, output_dim, batch_norm=False, dropout_rate=0.0, device=None):
        super().__init__()
        self.linear = rnn.Linear(input_dim, output_dim, device=device)
        self.bn = torch.nn.BatchNorm1d(output_dim, device=device) if batch_norm else torch.nn.Identity()
        self.act = torch.nn.ReLU()
        self.dropout = torch.nn.Dropout(p=dropout_rate)

    def forward(self, x):
        return self.dropout(self.act(self.bn(self.linear(x))))


class DCN(torch.nn.Module):
    """Deep & Cross Network over F categorical fields (DCN/main.py:72-190).  The F per-field tables are ONE table: field f, id i is
    row offsets[f] + i (`tables()` gives the reference's per-field views)."""

    def __init__(self, counts, embedding_dim=10, hidden_dims=(400, 400, 400), num_layers=3, batch_norm=False, hidden_dropout_rate=0.0,
                 device="cuda"):
        super().__init__()
        self.counts = list(counts)
        off = [0]
        for c in self.counts[:-1]:
            off.append(off[-1] + c)
        self.register_buffer("offsets", torch.tensor(off, dtype=torch.int64, device=device))
        self.embeddings = rnn.Embedding(sum(self.counts), embedding_dim, device=device)
        d_in = len(self.counts) * embedding_dim
        dims = [d_in] + list(hidden_dims)
        self.dnn = torch.nn.Sequential(*[MLPBlock(a, b, batch_norm, hidden_dropout_rate, device) for a, b in zip(dims[:-1], dims[1:])])
        self.crossnet = torch.nn.ModuleList([CrossInteraction(d_in, device) for _ in range(num_layers)])
        self.fc = rnn.Linear(d_in + dims[-1], 1, device=device)
        self.criterion = rnn.BCELoss4Logits(reduction="mean")
        with torch.no_grad():                                    # DCN.reset_parameters (DCN/main.py:122-131)
            torch.nn.init.normal_(self.embeddings.weight, std=1e-4)

    def tables(self):
        return [self.embeddings.weight[o:o + c] for o, c in zip(self.offsets.tolist(), self.counts)]

    def encode(self, x):
        """x [B, F] int64 field ids -> logits [B, 1]   (DCN/main.py:153-165)."""
        B = x.shape[0]
        emb = self.embeddings((x + self.offsets.unsqueeze(0)).reshape(-1)).reshape(B, -1)      # [B, F*D]
        deep = self.dnn(emb)
        cross = emb
        for layer in self.crossnet:
            cross = layer(emb, cross)
        return self.fc(torch.cat((deep, cross), dim=-1))

    def fit(self, x, labels):
        return {"rec_loss": self.criterion(self.encode(x), labels.to(torch.float32).reshape(-1, 1))}

    def recommend_from_pool(self, x):
        return self.encode(x).sigmoid()


# ------------------------------------------------------------------------------------------------ SimGCL
class SimGCL(_BufferRanking, torch.nn.Module):
    """SimGCL (SimGCL/main.py:34-160): LightGCN-style propagation with the layer-wise mean, BPR + L2 regulariser, and an InfoNCE
    loss between two noise-perturbed propagation views.  adj = (crow, col, val): the symmetric normalised bipartite adjacency as CSR."""

    def __init__(self, num_users, num_items, adj, embedding_dim=64, num_layers=3, eps=0.1, temperature=0.2, device="cuda"):
        super().__init__()
        self.U, self.N, self.num_layers, self.eps, self.temperature = num_users, num_items, num_layers, eps, temperature
        self.user = rnn.Embedding(num_users, embedding_dim, device=device)
        self.item = rnn.Embedding(num_items, embedding_dim, device=device)
        with torch.no_grad():
            torch.nn.init.xavier_uniform_(self.user.weight)
            torch.nn.init.xavier_uniform_(self.item.weight)
        crow, col, val = adj
        self.register_buffer("crow", crow.to(device)); self.register_buffer("col", col.to(device)); self.register_buffer("val", val.to(device))
        self.criterion = rnn.BPRLoss(reduction="mean")
        self.ranking_buffer = None

    def _propagate(self, noisy):
        x = torch.cat((self.user.weight, self.item.weight), dim=0)
        avg = 0.0
        for _ in range(self.num_layers):
            x = rnn.spmm_sym(self.crow, self.col, self.val, x)
            if noisy:
                x = x + self.eps * F.normalize(torch.rand_like(x), dim=-1).mul(x.sign())
            avg = avg + x / self.num_layers
        return torch.split(avg, (self.U, self.N))

    def encode(self):
        return self._propagate(False)

    def encode_(self):
        u, i = self._propagate(True)
        return F.normalize(u, dim=-1), F.normalize(i, dim=-1)

    def fit(self, users, positives, negatives):
        """users, positives, negatives: [B] int64 (one negative per positive, the reference's training pipe)."""
        users, positives, negatives = users.reshape(-1), positives.reshape(-1), negatives.reshape(-1)
        ue, ie = self.encode()
        rec_loss = rnn.bpr_triplet(ue.contiguous(), ie.contiguous(), users, positives, negatives)
        raw = (self.user(users), self.item(positives), self.item(negatives))
        emb_loss = sum(t.pow(2).sum() for t in raw) / 2 / users.numel()          # criterion.regularize(rtype="l2") / B
        u1, i1 = self.encode_()
        u2, i2 = self.encode_()
        targets = torch.arange(users.numel(), device=users.device)
        ssl = 0.0
        for a, b, idx in ((u1, u2, users), (i1, i2, positives)):
            logits = rnn.score_full(rnn.gather_rows(a.contiguous(), idx), rnn.gather_rows(b.contiguous(), idx)) / self.temperature
            ssl = ssl + F.cross_entropy(logits, targets)
        return {"rec_loss": rec_loss, "emb_loss": emb_loss, "ssl_loss": ssl}

    def reset_ranking_buffers(self):
        with torch.no_grad():
            ue, ie = self.encode()
            self.ranking_buffer = (ue.contiguous(), ie.contiguous())

    def recommend_from_full(self, users):
        ue, ie = self.ranking_buffer
        return rnn.score_full(rnn.gather_rows(ue, users.reshape(-1)), ie)


# ------------------------------------------------------------------------------------------------ GRU4Rec
class GRU4Rec(_EncodeRanking, torch.nn.Module):
    """GRU4Rec (GRU4Rec/main.py:30-190): item embeddings (row 0 = padding) -> dropout -> GRU -> dense -> the state at the last real
    position of the RIGHT-padded sequence -> pair / CE criterion against the item table."""

    def __init__(self, num_items, embedding_dim=64, hidden_size=128, num_blocks=1, emb_dropout_rate=0.2, hidden_dropout_rate=0.2,
                 loss="BCE", device="cuda"):
        super().__init__()
        assert loss in ("BCE", "BPR", "CE")
        self.N, self.loss_kind = num_items, loss
        self.item = rnn.Embedding(num_items + 1, embedding_dim, padding_idx=0, device=device)
        self.emb_dropout = torch.nn.Dropout(emb_dropout_rate)
        self.gru = torch.nn.GRU(embedding_dim, hidden_size, num_layers=num_blocks, bias=False, batch_first=True,
                                dropout=hidden_dropout_rate, device=device)
        self.dense = rnn.Linear(hidden_size, embedding_dim, device=device)
        with torch.no_grad():                                    # GRU4Rec.reset_parameters (GRU4Rec/main.py:76-82)
            torch.nn.init.xavier_normal_(self.item.weight)
            torch.nn.init.xavier_uniform_(self.gru.weight_hh_l0)
            torch.nn.init.xavier_uniform_(self.gru.weight_ih_l0)

    def encode(self, seqs):
        """seqs [B, S] int64, ids + 1, right-padded with 0 -> (userEmbds [B, D], itemEmbds [N, D])."""
        mask = seqs.ne(0)
        if not getattr(self, "static_shapes", False):
            # shrink_pads: columns that are padding for every row are dropped (GRU4Rec/main.py:121-124).  The state at a row's last real
            # position does not depend on the pad columns behind it, so keeping them (static_shapes: no data-dependent shape, no host
            # sync -- what a captured step needs) gives the same user states.
            keep = mask.any(dim=0)
            seqs, mask = seqs[:, keep], mask[:, keep]
        B, S = seqs.shape
        x = self.emb_dropout(self.item(seqs.reshape(-1)).reshape(B, S, -1))
        out, _ = self.gru(x)
        out = self.dense(out.reshape(B * S, -1)).reshape(B, S, -1)
        last = (mask.sum(1) - 1).clamp_min(0)
        user = rnn.gather_rows(out.reshape(B * S, -1), torch.arange(B, device=seqs.device) * S + last)
        return user, self.item.weight[1:]

    def fit(self, seqs, positives, negatives):
        """positives / negatives [B] int64 (0-based item ids): the last item as the target, one sampled negative."""
        user, items = self.encode(seqs)
        pos, neg = positives.reshape(-1), negatives.reshape(-1)
        if self.loss_kind == "BPR":
            ar = torch.arange(user.shape[0], device=user.device)
            return {"rec_loss": rnn.bpr_triplet(user.contiguous(), items.contiguous(), ar, pos, neg)}
        if self.loss_kind == "BCE":
            pl = (user * rnn.gather_rows(items.contiguous(), pos)).sum(-1, keepdim=True)
            nl = (user * rnn.gather_rows(items.contiguous(), neg)).sum(-1, keepdim=True)
            crit = rnn.BCELoss4Logits(reduction="mean")
            return {"rec_loss": crit(pl, torch.ones_like(pl)) + crit(nl, torch.zeros_like(nl))}
        return {"rec_loss": F.cross_entropy(rnn.score_full(user.contiguous(), items.contiguous()), pos)}

    def recommend_from_full(self, seqs):
        user, items = self.encode(seqs)
        return rnn.score_full(user.contiguous(), items.contiguous())


# ------------------------------------------------------------------------------------------------ NGCF
class NGCFConv(torch.nn.Module):
    """normalize(dropout(LeakyReLU(W1 (A x + x)) + LeakyReLU(W2 (A x * x))))   (NGCF/main.py:29-50)."""

    def __init__(self, in_features, out_features, dropout_rate=0.0, device=None):
        super().__init__()
        self.linear1 = rnn.Linear(in_features, out_features, device=device)
        self.linear2 = rnn.Linear(in_features, out_features, device=device)
        self.act = torch.nn.LeakyReLU()
        self.dropout = torch.nn.Dropout(p=dropout_rate)

    def forward(self, x, A, At):
        z = rnn.spmm(A, At, x)
        return F.normalize(self.dropout(self.act(self.linear1(z + x)) + self.act(self.linear2(z * x))), dim=-1)


class NGCF(_BufferRanking, torch.nn.Module):
    """NGCF (NGCF/main.py:53-160).  adj = (crow, col, val): D^-1 (A + I) of the bipartite interaction graph as CSR."""

    def __init__(self, num_users, num_items, adj, embedding_dim=64, num_layers=3, dropout_rate=0.0, device="cuda"):
        super().__init__()
        self.U, self.N = num_users, num_items
        self.user = rnn.Embedding(num_users, embedding_dim, device=device)
        self.item = rnn.Embedding(num_items, embedding_dim, device=device)
        with torch.no_grad():
            torch.nn.init.xavier_normal_(self.user.weight)
            torch.nn.init.xavier_normal_(self.item.weight)
        self.convs = torch.nn.ModuleList([NGCFConv(embedding_dim, embedding_dim, dropout_rate, device) for _ in range(num_layers)])
        A = tuple(t.to(device) for t in adj)
        At = rnn.csr_transpose(A, num_users + num_items)
        for n, t in zip(("crow", "col", "val", "t_crow", "t_col", "t_val"), A + At):
            self.register_buffer(n, t)
        self.ranking_buffer = None

    def encode(self):
        A, At = (self.crow, self.col, self.val), (self.t_crow, self.t_col, self.t_val)
        x = torch.cat((self.user.weight, self.item.weight), dim=0)
        outs = [x]
        for conv in self.convs:
            x = conv(x, A, At)
            outs.append(x)
        return torch.split(torch.cat(outs, dim=-1), (self.U, self.N))

    def fit(self, users, positives, negatives):
        users, positives, negatives = users.reshape(-1), positives.reshape(-1), negatives.reshape(-1)
        ue, ie = self.encode()
        rec_loss = rnn.bpr_triplet(ue.contiguous(), ie.contiguous(), users, positives, negatives)
        raw = (self.user(users), self.item(positives), self.item(negatives))
        return {"rec_loss": rec_loss, "emb_loss": sum(t.pow(2).sum() for t in raw) / 2 / users.numel()}

    def reset_ranking_buffers(self):
        with torch.no_grad():
            ue, ie = self.encode()
            self.ranking_buffer = (ue.contiguous(), ie.contiguous())

    def recommend_from_full(self, users):
        ue, ie = self.ranking_buffer
        return rnn.linear(rnn.gather_rows(ue, users.reshape(-1)), ie)


# ------------------------------------------------------------------------------------------------ SGL
class SGL(_BufferRanking, torch.nn.Module):
    """SGL (SGL/main.py:30-215).  edges = (users [E], items [E]): the training interactions.  The undirected bipartite graph's CSR
    pattern is built once; `resample()` (the reference's Coach calls it at the top of every epoch, SGL/main.py:245-246) draws the two
    subgraphs' edge weights and re-normalises D^-1/2 A_w D^-1/2 on that pattern."""

    def __init__(self, num_users, num_items, edges, embedding_dim=64, num_layers=3, aug_type="ed", ssl_drop_rate=0.1, temperature=0.2,
                 device="cuda"):
        super().__init__()
        assert aug_type in ("nd", "ed", "rw")
        self.U, self.N, self.num_layers, self.aug_type, self.rate, self.temperature = num_users, num_items, num_layers, aug_type, ssl_drop_rate, temperature
        self.user = rnn.Embedding(num_users, embedding_dim, device=device)
        self.item = rnn.Embedding(num_items, embedding_dim, device=device)
        with torch.no_grad():
            torch.nn.init.xavier_uniform_(self.user.weight)
            torch.nn.init.xavier_uniform_(self.item.weight)
        eu, ei = (torch.as_tensor(t, dtype=torch.int64).reshape(-1) for t in edges)
        E, n = eu.numel(), num_users + num_items
        rows = torch.cat((eu, ei + num_users)); cols = torch.cat((ei + num_users, eu))
        eid = torch.cat((torch.arange(E), torch.arange(E)))                  # CSR entry -> the interaction it comes from
        order = torch.argsort(rows * n + cols)
        crow = torch.zeros(n + 1, dtype=torch.int64)
        crow[1:] = torch.cumsum(torch.bincount(rows, minlength=n), 0)
        for name, t in (("crow", crow), ("col", cols[order].contiguous()), ("row", rows[order].contiguous()), ("eid", eid[order].contiguous()),
                        ("edge_u", eu), ("edge_i", ei)):
            self.register_buffer(name, t.to(device))
        self.register_buffer("val", self._normalised(torch.ones(E, device=device)))
        self.sub = self.sub_ = None
        self.ranking_buffer = None

    def _normalised(self, edge_weight):
        """Per-interaction weights [E] -> the CSR values of D^-1/2 A_w D^-1/2 (a node left without edges keeps zero entries)."""
        w = edge_weight[self.eid]
        deg = torch.zeros(self.U + self.N, device=w.device).index_add_(0, self.row, w)
        dinv = torch.where(deg > 0, deg.rsqrt(), torch.zeros_like(deg))
        return (w * dinv[self.row] * dinv[self.col]).contiguous()

    def sample_subgraph(self, rnd=None):
        """SGL.sample_subgraph (SGL/main.py:87-111); rnd: the uniform draws to use instead of torch.rand (tests) -- a [E] tensor for
        "ed" / "rw", a ([U], [N]) pair for "nd"."""
        dev = self.val.device
        if self.aug_type == "nd":
            ru, ri = rnd if rnd is not None else (torch.rand(self.U, device=dev), torch.rand(self.N, device=dev))
            w = (ru > self.rate).float()[self.edge_u] * (ri > self.rate).float()[self.edge_i]
        else:
            r = rnd if rnd is not None else torch.rand(self.edge_u.numel(), device=dev)
            w = (r > self.rate).float()
        return self._normalised(w)

    def resample(self, rnds=None):
        """Two views; "nd" / "ed": one subgraph per view shared by all layers, "rw": one per layer (SGL/main.py:113-121)."""
        k = 1 if self.aug_type in ("nd", "ed") else self.num_layers
        it = iter(rnds) if rnds is not None else None
        nxt = (lambda: next(it)) if it is not None else (lambda: None)
        a = [self.sample_subgraph(nxt()) for _ in range(k)]
        b = [self.sample_subgraph(nxt()) for _ in range(k)]
        self.sub, self.sub_ = a * (self.num_layers // k), b * (self.num_layers // k)

    def _propagate(self, vals):
        x = torch.cat((self.user.weight, self.item.weight), dim=0)
        avg = x / (self.num_layers + 1)
        for l in range(self.num_layers):
            x = rnn.spmm_sym(self.crow, self.col, vals[l], x.contiguous())
            avg = avg + x / (self.num_layers + 1)
        return torch.split(avg, (self.U, self.N))

    def encode(self):
        return self._propagate([self.val] * self.num_layers)

    def fit(self, users, positives, negatives):
        users, positives, negatives = users.reshape(-1), positives.reshape(-1), negatives.reshape(-1)
        if self.sub is None:
            self.resample()
        ue, ie = self.encode()
        rec_loss = rnn.bpr_triplet(ue.contiguous(), ie.contiguous(), users, positives, negatives)
        raw = (self.user(users), self.item(positives), self.item(negatives))
        emb_loss = sum(t.pow(2).sum() for t in raw) / 2 / users.numel()
        (u1, i1), (u2, i2) = self._propagate(self.sub), self._propagate(self.sub_)
        targets = torch.arange(users.numel(), device=users.device)
        ssl = 0.0
        for a, b, idx in ((u1, u2, users), (i1, i2, positives)):
            a, b = F.normalize(a, dim=-1), F.normalize(b, dim=-1)
            logits = rnn.score_full(rnn.gather_rows(a.contiguous(), idx), rnn.gather_rows(b.contiguous(), idx)) / self.temperature
            ssl = ssl + F.cross_entropy(logits, targets)
        return {"rec_loss": rec_loss, "emb_loss": emb_loss, "ssl_loss": ssl}

    def reset_ranking_buffers(self):
        with torch.no_grad():
            ue, ie = self.encode()
            self.ranking_buffer = (ue.contiguous(), ie.contiguous())

    def recommend_from_full(self, users):
        ue, ie = self.ranking_buffer
        return rnn.score_full(rnn.gather_rows(ue, users.reshape(-1)), ie)


# ------------------------------------------------------------------------------------------------ JGCF
class JGCF(_BufferRanking, torch.nn.Module):
    """JGCF (JGCF/main.py:39-160): z_0 = X, z_l from the Jacobi three-term recurrence on the symmetric normalised adjacency
    (JGCF/modules.py:8-49), low = mean_l(coef_l z_l) with coef = cumprod(tanh(gamma) * scaling) (modules.py:77-83; gamma is a frozen
    parameter), mid = weight4mid * X - low, tables = [low | mid]; BPR + L2 regulariser."""

    def __init__(self, num_users, num_items, adj, embedding_dim=64, num_layers=3, scaling_factor=3.0, alpha=1.0, beta=1.0, weight4mid=0.1,
                 device="cuda"):
        super().__init__()
        self.U, self.N, self.L, self.scaling, self.a, self.b, self.weight4mid = num_users, num_items, num_layers, scaling_factor, alpha, beta, weight4mid
        self.user = rnn.Embedding(num_users, embedding_dim, device=device)
        self.item = rnn.Embedding(num_items, embedding_dim, device=device)
        with torch.no_grad():
            torch.nn.init.normal_(self.user.weight, std=1e-4)
            torch.nn.init.normal_(self.item.weight, std=1e-4)
        self.gammas = torch.nn.Parameter(torch.full((num_layers + 1, 1), min(1.0 / scaling_factor, 1.0), device=device), requires_grad=False)
        for n, t in zip(("crow", "col", "val"), adj):
            self.register_buffer(n, t.to(device))
        self.ranking_buffer = None

    def _conv(self, x):
        a, b = self.a, self.b
        zs = [x]
        for l in range(1, self.L + 1):
            Az = rnn.spmm_sym(self.crow, self.col, self.val, zs[-1].contiguous())
            if l == 1:
                z = (a - b) / 2 * zs[-1] + (a + b + 2) / 2 * Az
            else:
                c0 = 2 * l * (l + a + b) * (2 * l + a + b - 2)
                c1 = (2 * l + a + b - 1) * (a ** 2 - b ** 2)
                c2 = (2 * l + a + b - 1) * (2 * l + a + b) * (2 * l + a + b - 2)
                c3 = 2 * (l + a - 1) * (l + b - 1) * (2 * l + a + b)
                z = (c1 * zs[-1] + c2 * Az - c3 * zs[-2]) / c0
            zs.append(z)
        coefs = (self.gammas.tanh() * self.scaling).cumprod(dim=0)
        return (torch.stack(zs, dim=1) * coefs).mean(1)

    def encode(self):
        x = torch.cat((self.user.weight, self.item.weight), dim=0)
        low = self._conv(x)
        return torch.split(torch.cat((low, self.weight4mid * x - low), dim=1), (self.U, self.N))

    def fit(self, users, positives, negatives):
        users, positives, negatives = users.reshape(-1), positives.reshape(-1), negatives.reshape(-1)
        ue, ie = self.encode()
        rec_loss = rnn.bpr_triplet(ue.contiguous(), ie.contiguous(), users, positives, negatives)
        raw = (self.user(users), self.item(positives), self.item(negatives))
        return {"rec_loss": rec_loss, "emb_loss": sum(t.pow(2).sum() for t in raw) / 2 / users.numel()}

    def reset_ranking_buffers(self):
        with torch.no_grad():
            ue, ie = self.encode()
            self.ranking_buffer = (ue.contiguous(), ie.contiguous())

    def recommend_from_full(self, users):
        ue, ie = self.ranking_buffer
        return rnn.linear(rnn.gather_rows(ue, users.reshape(-1)), ie)


# ------------------------------------------------------------------------------------------------ BERT4Rec
class BERT4Rec(torch.nn.Module):
    """BERT4Rec (BERT4Rec/main.py:33-195): item + position embeddings -> LayerNorm -> dropout -> bidirectional TransformerEncoder
    (post-norm, GELU, feed-forward 4 D) -> `fc` to N + NUM_PADS logits -> cross entropy at the masked positions.
    Ids: 0 = padding, 1 = the mask token, item i = i + 2."""

    NUM_PADS, PADDING_VALUE, MASKING_VALUE = 2, 0, 1

    def __init__(self, num_items, maxlen=50, embedding_dim=64, num_heads=4, num_blocks=2, mask_ratio=0.3, dropout_rate=0.2, device="cuda"):
        super().__init__()
        self.N, self.maxlen, self.mask_ratio = num_items, maxlen, mask_ratio
        self.item = rnn.Embedding(num_items + self.NUM_PADS, embedding_dim, padding_idx=self.PADDING_VALUE, device=device)
        self.Position = torch.nn.Embedding(maxlen, embedding_dim, device=device)
        self.layernorm = torch.nn.LayerNorm(embedding_dim, device=device)
        self.dropout = torch.nn.Dropout(dropout_rate)
        self.encoder = torch.nn.TransformerEncoder(
            torch.nn.TransformerEncoderLayer(d_model=embedding_dim, nhead=num_heads, dim_feedforward=embedding_dim * 4, dropout=dropout_rate,
                                             activation="gelu", batch_first=True, device=device), num_layers=num_blocks)
        self.fc = rnn.Linear(embedding_dim, num_items + self.NUM_PADS, device=device)
        with torch.no_grad():                                    # BERT4Rec.reset_parameters (BERT4Rec/main.py:89-98)
            for m in self.modules():
                if isinstance(m, (torch.nn.Linear, rnn.Linear, torch.nn.Embedding, rnn.Embedding)):
                    torch.nn.init.xavier_normal_(m.weight)
                    m.weight.clamp_(-0.02, 0.02)
                    if getattr(m, "bias", None) is not None:
                        m.bias.zero_()

    def random_mask(self, seqs, p, rnds=None):
        """BERT4Rec.random_mask (BERT4Rec/main.py:155-164); `rnds` [B, S] in [0, 1) stands in for torch.rand (tests)."""
        pad = seqs == self.PADDING_VALUE
        if rnds is None:
            rnds = torch.rand(seqs.shape, device=seqs.device)
        masked = torch.where(rnds < p, torch.full_like(seqs, self.MASKING_VALUE), seqs).masked_fill(pad, self.PADDING_VALUE)
        masks = masked == self.MASKING_VALUE
        return masked, seqs[masks], masks

    def encode(self, seqs):
        """seqs [B, maxlen] int64 -> states [B, maxlen, D]."""
        B, S = seqs.shape
        pad = seqs == self.PADDING_VALUE
        x = self.item(seqs.reshape(-1)).reshape(B, S, -1) + self.Position.weight[:S].unsqueeze(0)
        x = self.dropout(self.layernorm(x))
        return self.encoder(x, src_key_padding_mask=pad)

    def fit(self, seqs, rnds=None):
        if getattr(self, "static_shapes", False):
            return self._fit_static(seqs, rnds)
        masked, labels, masks = self.random_mask(seqs, self.mask_ratio, rnds)
        h = self.encode(masked)
        rows = rnn.gather_rows(h.reshape(-1, h.shape[-1]).contiguous(), masks.reshape(-1).nonzero().squeeze(1))
        return {"rec_loss": F.cross_entropy(self.fc(rows), labels)}

    def _fit_static(self, seqs, rnds=None):
        """The same loss with no data-dependent shape (no host sync: a captured step).  The masked positions are compacted into a buffer
        of FIXED capacity (torch.nonzero_static): the mean over n p masked positions plus 8 standard deviations of the binomial count
        (the count exceeds it with probability < 1e-15; positions beyond it would be left out of the mean), unused slots are zero rows
        whose label is cross_entropy's ignore_index."""
        pad = seqs == self.PADDING_VALUE
        if rnds is None:
            rnds = torch.rand(seqs.shape, device=seqs.device)
        masked = torch.where(rnds < self.mask_ratio, torch.full_like(seqs, self.MASKING_VALUE), seqs).masked_fill(pad, self.PADDING_VALUE)
        masks = (masked == self.MASKING_VALUE).reshape(-1)
        n, p = masks.numel(), self.mask_ratio
        cap = min(n, int(n * p + 8.0 * (n * p * (1.0 - p)) ** 0.5) + 16)
        at = torch.nonzero_static(masks, size=cap, fill_value=-1).squeeze(1)
        h = self.encode(masked)
        rows = rnn.gather_rows(h.reshape(-1, h.shape[-1]).contiguous(), at)                    # (-1: a zero row)
        labels = torch.where(at >= 0, seqs.reshape(-1)[at.clamp_min(0)], torch.full_like(at, -100))
        return {"rec_loss": F.cross_entropy(self.fc(rows), labels, ignore_index=-100)}

    def recommend_from_full(self, seqs):
        """seqs: the evaluation pipe's rows (left-padded history of maxlen - 1, the mask token last)."""
        return self.fc(self.encode(seqs)[:, -1, :].contiguous())[:, self.NUM_PADS:]


# ------------------------------------------------------------------------------------------------ shared: last-item criteria
def _last_item_loss(kind, user, items, pos, neg):
    """BCE / BPR / CE of one user state per sequence against the item table (GRU4Rec/main.py:156-180, STAMP/main.py:128-150,
    FMLP-Rec/main.py:152-174)."""
    pos, neg = pos.reshape(-1), neg.reshape(-1)
    if kind == "BPR":
        return rnn.bpr_triplet(user.contiguous(), items.contiguous(), torch.arange(user.shape[0], device=user.device), pos, neg)
    if kind == "BCE":
        pl = (user * rnn.gather_rows(items.contiguous(), pos)).sum(-1, keepdim=True)
        nl = (user * rnn.gather_rows(items.contiguous(), neg)).sum(-1, keepdim=True)
        crit = rnn.BCELoss4Logits(reduction="mean")
        return crit(pl, torch.ones_like(pl)) + crit(nl, torch.zeros_like(nl))
    return F.cross_entropy(rnn.score_full(user.contiguous(), items.contiguous()), pos)


def _lin3(layer, x):
    """A 2-D engine Linear over the last dimension of a [B, S, D] tensor."""
    B, S, D = x.shape
    return layer(x.reshape(B * S, D)).reshape(B, S, -1)


# ------------------------------------------------------------------------------------------------ GCN
class GCN(_BufferRanking, torch.nn.Module):
    """GCN (GCN/main.py:27-135): x <- Linear_l(Adj x), ReLU after all but the last layer; BPR on the propagated tables."""

    def __init__(self, num_users, num_items, adj, embedding_dim=64, num_layers=3, device="cuda"):
        super().__init__()
        self.U, self.N = num_users, num_items
        self.user = rnn.Embedding(num_users, embedding_dim, device=device)
        self.item = rnn.Embedding(num_items, embedding_dim, device=device)
        with torch.no_grad():
            torch.nn.init.normal_(self.user.weight, std=1e-4)
            torch.nn.init.normal_(self.item.weight, std=1e-4)
        self.linears = torch.nn.ModuleList([rnn.Linear(embedding_dim, embedding_dim, device=device) for _ in range(num_layers)])
        for n, t in zip(("crow", "col", "val"), adj):
            self.register_buffer(n, t.to(device))
        self.ranking_buffer = None

    def encode(self):
        x = torch.cat((self.user.weight, self.item.weight), dim=0)
        for l, lin in enumerate(self.linears):
            x = lin(rnn.spmm_sym(self.crow, self.col, self.val, x.contiguous()))
            if l + 1 < len(self.linears):
                x = torch.relu(x)
        return torch.split(x, (self.U, self.N))

    def fit(self, users, positives, negatives):
        ue, ie = self.encode()
        return {"rec_loss": rnn.bpr_triplet(ue.contiguous(), ie.contiguous(), users.reshape(-1), positives.reshape(-1), negatives.reshape(-1))}

    def reset_ranking_buffers(self):
        with torch.no_grad():
            ue, ie = self.encode()
            self.ranking_buffer = (ue.contiguous(), ie.contiguous())

    def recommend_from_full(self, users):
        ue, ie = self.ranking_buffer
        return rnn.linear(rnn.gather_rows(ue, users.reshape(-1)), ie)


# ------------------------------------------------------------------------------------------------ STAMP
class STAMP(_EncodeRanking, torch.nn.Module):
    """STAMP (STAMP/main.py:28-160): mean of the sequence's embeddings + the last click -> trilinear attention -> two tanh MLP branches,
    multiplied.  Sequences are LEFT-padded (the last column is the last click); ids + 1, 0 = padding."""

    def __init__(self, num_items, embedding_dim=64, hidden_size=64, loss="BCE", device="cuda"):
        super().__init__()
        assert loss in ("BCE", "BPR", "CE")
        D = embedding_dim
        self.N, self.loss_kind = num_items, loss
        self.item = rnn.Embedding(num_items + 1, D, padding_idx=0, device=device)
        self.w1, self.w2, self.w3 = (rnn.Linear(D, D, bias=False, device=device) for _ in range(3))
        self.w0 = rnn.Linear(D, 1, bias=False, device=device)
        self.ba = torch.nn.Parameter(torch.zeros(1, 1, D, device=device))
        self.mlp_a = rnn.Linear(D, hidden_size, device=device)
        self.mlp_b = rnn.Linear(D, hidden_size, device=device)
        with torch.no_grad():                                    # STAMP.reset_parameters (STAMP/main.py:78-84)
            for m in (self.w0, self.w1, self.w2, self.w3, self.mlp_a, self.mlp_b):
                torch.nn.init.normal_(m.weight, std=0.05)
            torch.nn.init.normal_(self.item.weight, std=0.002)

    def encode(self, seqs):
        B, S = seqs.shape
        lens = seqs.ne(0).sum(dim=-1, keepdim=True)
        x = self.item(seqs.reshape(-1)).reshape(B, S, -1)
        last = x[:, -1, :].contiguous()
        ms = x.sum(dim=1).div(lens)
        alphas = _lin3(self.w0, torch.sigmoid(_lin3(self.w1, x) + self.w2(last).unsqueeze(1) + self.w3(ms).unsqueeze(1) + self.ba))
        ma = alphas.mul(x).sum(1) + last
        return torch.tanh(self.mlp_a(ma)) * torch.tanh(self.mlp_b(last)), self.item.weight[1:]

    def fit(self, seqs, positives, negatives):
        user, items = self.encode(seqs)
        return {"rec_loss": _last_item_loss(self.loss_kind, user, items, positives, negatives)}

    def recommend_from_full(self, seqs):
        user, items = self.encode(seqs)
        return rnn.score_full(user.contiguous(), items.contiguous())


# ------------------------------------------------------------------------------------------------ NARM
class NARM(_EncodeRanking, torch.nn.Module):
    """NARM (NARM/main.py:30-180): GRU over the RIGHT-padded sequence; global = the state at the last real position, local = attention
    over the states (v_t(mask * sigmoid(a_1 h_s + a_2 h_t))); user = b [local | global]; BCE."""

    def __init__(self, num_items, embedding_dim=64, hidden_size=128, num_blocks=1, emb_dropout_rate=0.25, hidden_dropout_rate=0.0,
                 ct_dropout_rate=0.5, device="cuda"):
        super().__init__()
        H = hidden_size
        self.N = num_items
        self.item = rnn.Embedding(num_items + 1, embedding_dim, padding_idx=0, device=device)
        self.emb_dropout = torch.nn.Dropout(emb_dropout_rate)
        self.gru = torch.nn.GRU(embedding_dim, H, num_layers=num_blocks, bias=False, batch_first=True, dropout=hidden_dropout_rate, device=device)
        self.a_1 = rnn.Linear(H, H, bias=False, device=device)
        self.a_2 = rnn.Linear(H, H, bias=False, device=device)
        self.v_t = rnn.Linear(H, 1, bias=False, device=device)
        self.ct_dropout = torch.nn.Dropout(ct_dropout_rate)
        self.b = rnn.Linear(2 * H, embedding_dim, bias=False, device=device)
        with torch.no_grad():
            torch.nn.init.xavier_normal_(self.item.weight)

    def encode(self, seqs):
        mask = seqs.ne(0)
        if not getattr(self, "static_shapes", False):            # shrink_pads (NARM/main.py:131-134); static_shapes: as in GRU4Rec.encode -- the
            keep = mask.any(dim=0)                               # pad columns carry mask 0 into the attention, the result is the same
            seqs, mask = seqs[:, keep], mask[:, keep]
        B, S = seqs.shape
        out, _ = self.gru(self.emb_dropout(self.item(seqs.reshape(-1)).reshape(B, S, -1)))
        last = (mask.sum(1) - 1).clamp_min(0)
        ht = rnn.gather_rows(out.reshape(B * S, -1).contiguous(), torch.arange(B, device=seqs.device) * S + last)          # [B, H]
        alpha = _lin3(self.v_t, mask.unsqueeze(-1) * torch.sigmoid(_lin3(self.a_1, out) + self.a_2(ht).unsqueeze(1)))
        c_t = self.ct_dropout(torch.cat([(alpha * out).sum(1), ht], 1))
        return self.b(c_t), self.item.weight[1:]

    def fit(self, seqs, positives, negatives):
        user, items = self.encode(seqs)
        return {"rec_loss": _last_item_loss("BCE", user, items, positives, negatives)}

    def recommend_from_full(self, seqs):
        user, items = self.encode(seqs)
        return rnn.score_full(user.contiguous(), items.contiguous())


# ------------------------------------------------------------------------------------------------ FMLP-Rec
class _TFLayerNorm(torch.nn.Module):
    """LayerNorm with epsilon inside the square root (FMLP-Rec/modules.py:27-40)."""

    def __init__(self, size, eps=1e-12, device=None):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.ones(size, device=device))
        self.bias = torch.nn.Parameter(torch.zeros(size, device=device))
        self.eps = eps

    def forward(self, x):
        u = x.mean(-1, keepdim=True)
        s = (x - u).pow(2).mean(-1, keepdim=True)
        return self.weight * ((x - u) / torch.sqrt(s + self.eps)) + self.bias


class _FilterBlock(torch.nn.Module):
    """FilterLayer + Intermediate (FMLP-Rec/modules.py:42-96): rfft over the sequence, a learnable complex filter, irfft, residual
    LayerNorm; then dense_1 -> GELU -> dense_2, residual LayerNorm."""

    def __init__(self, maxlen, D, dropout_rate, device):
        super().__init__()
        self.complex_weight = torch.nn.Parameter(torch.randn(1, maxlen // 2 + 1, D, 2, device=device) * 0.02)
        self.filter_norm = _TFLayerNorm(D, device=device)
        self.dense_1 = rnn.Linear(D, 4 * D, device=device)
        self.dense_2 = rnn.Linear(4 * D, D, device=device)
        self.out_norm = _TFLayerNorm(D, device=device)
        self.dropout = torch.nn.Dropout(dropout_rate)

    def forward(self, x):
        S = x.shape[1]
        f = torch.fft.irfft(torch.fft.rfft(x, dim=1, norm="ortho") * torch.view_as_complex(self.complex_weight), n=S, dim=1, norm="ortho")
        h = self.filter_norm(self.dropout(f) + x)
        z = _lin3(self.dense_1, h)
        z = z * 0.5 * (1.0 + torch.erf(z / 2.0 ** 0.5))
        return self.out_norm(self.dropout(_lin3(self.dense_2, z)) + h)


class FMLPRec(_EncodeRanking, torch.nn.Module):
    """FMLP-Rec (FMLP-Rec/main.py:38-180): item + position embeddings -> LayerNorm -> dropout -> filter-enhanced blocks -> the state at
    the last position of the LEFT-padded sequence -> BPR / BCE / CE against the item table."""

    def __init__(self, num_items, maxlen=50, embedding_dim=64, num_blocks=2, hidden_dropout_rate=0.5, loss="BPR", device="cuda"):
        super().__init__()
        assert loss in ("BCE", "BPR", "CE")
        D = embedding_dim
        self.N, self.loss_kind = num_items, loss
        self.item = rnn.Embedding(num_items + 1, D, padding_idx=0, device=device)
        self.Position = torch.nn.Embedding(maxlen, D, device=device)
        self.layerNorm = _TFLayerNorm(D, device=device)
        self.embdDropout = torch.nn.Dropout(hidden_dropout_rate)
        self.blocks = torch.nn.ModuleList([_FilterBlock(maxlen, D, hidden_dropout_rate, device) for _ in range(num_blocks)])
        with torch.no_grad():                                    # FMLPRec.reset_parameters (FMLP-Rec/main.py:80-90)
            for m in self.modules():
                if isinstance(m, (rnn.Linear, rnn.Embedding, torch.nn.Embedding)):
                    torch.nn.init.normal_(m.weight, std=0.02)
                    if getattr(m, "bias", None) is not None:
                        m.bias.zero_()

    def encode(self, seqs):
        B, S = seqs.shape
        x = self.item(seqs.reshape(-1)).reshape(B, S, -1) + self.Position.weight[:S].unsqueeze(0)
        x = self.embdDropout(self.layerNorm(x))
        for blk in self.blocks:
            x = blk(x)
        return x[:, -1, :].contiguous(), self.item.weight[1:]

    def fit(self, seqs, positives, negatives):
        user, items = self.encode(seqs)
        return {"rec_loss": _last_item_loss(self.loss_kind, user, items, positives, negatives)}

    def recommend_from_full(self, seqs):
        user, items = self.encode(seqs)
        return rnn.score_full(user.contiguous(), items.contiguous())


# ------------------------------------------------------------------------------------------------ BSARec
class _BSAFrequency(torch.nn.Module):
    """FrequencyLayer (BSARec/modules.py:146-171): low-pass = the first c // 2 + 1 rfft bins, high-pass = the rest re-weighted by sqrt_beta^2."""

    def __init__(self, D, c, dropout_rate, device):
        super().__init__()
        self.c = c // 2 + 1
        self.sqrt_beta = torch.nn.Parameter(torch.randn(1, 1, D, device=device))
        self.LayerNorm = _TFLayerNorm(D, device=device)
        self.out_dropout = torch.nn.Dropout(dropout_rate)

    def forward(self, x):
        S = x.shape[1]
        f = torch.fft.rfft(x, dim=1, norm="ortho")
        f = torch.cat((f[:, :self.c], torch.zeros_like(f[:, self.c:])), dim=1)
        low = torch.fft.irfft(f, n=S, dim=1, norm="ortho")
        return self.LayerNorm(self.out_dropout(low + self.sqrt_beta ** 2 * (x - low)) + x)


class _BSAAttention(torch.nn.Module):
    """MultiHeadAttention (BSARec/modules.py:83-143): additive mask (0 / -1e4), softmax, residual nn.LayerNorm(eps = 1e-12)."""

    def __init__(self, D, num_heads, attn_dropout_rate, hidden_dropout_rate, device):
        super().__init__()
        self.H, self.dh = num_heads, D // num_heads
        self.query, self.key, self.value, self.dense = (rnn.Linear(D, D, device=device) for _ in range(4))
        self.LayerNorm = torch.nn.LayerNorm(D, eps=1e-12, device=device)
        self.attn_dropout, self.out_dropout = torch.nn.Dropout(attn_dropout_rate), torch.nn.Dropout(hidden_dropout_rate)

    def forward(self, x, mask):
        B, S, D = x.shape
        q, k, v = (_lin3(m, x).view(B, S, self.H, self.dh).permute(0, 2, 1, 3) for m in (self.query, self.key, self.value))
        p = self.attn_dropout(torch.softmax(q @ k.transpose(-1, -2) / self.dh ** 0.5 + mask, dim=-1))
        ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, S, D)
        return self.LayerNorm(self.out_dropout(_lin3(self.dense, ctx)) + x)


class _BSALayer(torch.nn.Module):
    def __init__(self, D, num_heads, c, alpha, attn_dropout_rate, hidden_dropout_rate, device):
        super().__init__()
        self.alpha = alpha
        self.filter_layer = _BSAFrequency(D, c, hidden_dropout_rate, device)
        self.attention_layer = _BSAAttention(D, num_heads, attn_dropout_rate, hidden_dropout_rate, device)

    def forward(self, x, mask):
        return self.alpha * self.filter_layer(x) + (1 - self.alpha) * self.attention_layer(x, mask)


class _BSAFeedForward(torch.nn.Module):
    def __init__(self, D, dropout_rate, device):
        super().__init__()
        self.dense_1, self.dense_2 = rnn.Linear(D, 4 * D, device=device), rnn.Linear(4 * D, D, device=device)
        self.LayerNorm = _TFLayerNorm(D, device=device)
        self.dropout = torch.nn.Dropout(dropout_rate)

    def forward(self, x):
        z = _lin3(self.dense_1, x)
        z = z * 0.5 * (1.0 + torch.erf(z / 2.0 ** 0.5))
        return self.LayerNorm(self.dropout(_lin3(self.dense_2, z)) + x)


class _BSABlock(torch.nn.Module):
    def __init__(self, D, num_heads, c, alpha, attn_dropout_rate, hidden_dropout_rate, device):
        super().__init__()
        self.layer = _BSALayer(D, num_heads, c, alpha, attn_dropout_rate, hidden_dropout_rate, device)
        self.feed_forward = _BSAFeedForward(D, hidden_dropout_rate, device)

    def forward(self, x, mask):
        return self.feed_forward(self.layer(x, mask))


class BSARec(_EncodeRanking, torch.nn.Module):
    """BSARec (BSARec/main.py:38-200): per block alpha * (frequency-domain low-pass branch) + (1 - alpha) * (causal self-attention), then
    the feed-forward; the state at the last position of the LEFT-padded sequence against the item table (BCE / BPR / CE)."""

    def __init__(self, num_items, maxlen=50, embedding_dim=64, num_heads=1, num_blocks=2, c=5, alpha=0.7, hidden_dropout_rate=0.5,
                 attn_dropout_rate=0.5, loss="CE", device="cuda"):
        super().__init__()
        assert loss in ("BCE", "BPR", "CE")
        D = embedding_dim
        self.N, self.loss_kind = num_items, loss
        self.item = rnn.Embedding(num_items + 1, D, padding_idx=0, device=device)
        self.Position = torch.nn.Embedding(maxlen, D, device=device)
        self.layerNorm = torch.nn.LayerNorm(D, eps=1e-12, device=device)
        self.embdDropout = torch.nn.Dropout(hidden_dropout_rate)
        self.blocks = torch.nn.ModuleList([_BSABlock(D, num_heads, c, alpha, attn_dropout_rate, hidden_dropout_rate, device) for _ in range(num_blocks)])
        with torch.no_grad():                                    # BSARec.reset_parameters (BSARec/main.py:88-100)
            for m in self.modules():
                if isinstance(m, (rnn.Embedding, torch.nn.Embedding)):
                    m.weight.normal_(mean=0.0, std=0.02)

    def encode(self, seqs):
        B, S = seqs.shape
        keep = (seqs != 0).view(B, 1, 1, S).expand(-1, -1, S, -1).tril()
        mask = torch.where(keep, 0.0, -1.0e4)
        x = self.item(seqs.reshape(-1)).reshape(B, S, -1) + self.Position.weight[:S].unsqueeze(0)
        x = self.embdDropout(self.layerNorm(x))
        for blk in self.blocks:
            x = blk(x, mask)
        return x[:, -1, :].contiguous(), self.item.weight[1:]

    def fit(self, seqs, positives, negatives):
        user, items = self.encode(seqs)
        return {"rec_loss": _last_item_loss(self.loss_kind, user, items, positives, negatives)}

    def recommend_from_full(self, seqs):
        user, items = self.encode(seqs)
        return rnn.score_full(user.contiguous(), items.contiguous())
