"""DeepFM on the engine: host-side mirror of the reference's `DeepFM` (DeepFM/main.py:127-219) and of
`CoachForDeepFM.train_per_epoch`'s step body (:256-276).

Native (librecengine): the multi-field embedding bag, the FM second-order term, the logistic-regression term, the
BCE-with-logits criterion and its gradient, the scatter-add of ALL field gradients (one launch per table) and the Adam
update of the tables.  The F per-field tables are ONE concatenated table (plus one LR vector): `field f, id i` is row
`offsets[f] + i`.  State-dict access keeps the reference's per-field view (`tables[f]`, `tables_lr[f]`).

STILL ON ATEN THIS ROUND (flagged in DESIGN.md §7): the 3x400 MLP with BatchNorm (DeepFM/main.py:103-124,151-164) and
its optimizer.  Its input gradient flows back into the native bag backward.
"""
import torch
import torch.nn as nn

from . import ops


class MLPBlock(nn.Module):
    """Linear -> BatchNorm1d -> ReLU -> Dropout (DeepFM/main.py:103-124)."""

    def __init__(self, i, o, batch_norm, p):
        super().__init__()
        self.linear = nn.Linear(i, o)
        self.bn = nn.BatchNorm1d(o) if batch_norm else nn.Identity()
        self.act = nn.ReLU()
        self.dropout = nn.Dropout(p)

    def forward(self, x):
        return self.dropout(self.act(self.bn(self.linear(x))))


class DeepFMEngine:
    def __init__(self, counts, embedding_dim=10, hidden_dims=(400, 400, 400), batch_norm=True, hidden_dropout_rate=0.0,
                 lr=1e-3, embedding_decay=0.05, weight_decay=0.0, betas=(0.9, 0.999), device="cuda", seed=1):
        self.counts, self.F, self.D = list(counts), len(counts), embedding_dim
        self.device = torch.device(device)
        self.lr, self.emb_decay, self.wd, self.betas = lr, embedding_decay, weight_decay, betas
        off = [0]
        for c in self.counts[:-1]:
            off.append(off[-1] + c)
        self.rows = sum(self.counts)
        self.offsets = torch.tensor(off, dtype=torch.int64, device=self.device)
        # arena: [T (rows*D) | TL (rows) | lr bias (1)]  -> one Adam launch with weight_decay = embedding_decay.
        # (the LR bias belongs to the reference's non-embedding group; its decay is applied separately below)
        nT, nL = self.rows * self.D, self.rows
        pad = (-(nT + nL)) % 4
        self.n_arena = nT + nL + pad + 4
        self.data = torch.zeros(self.n_arena, device=self.device)
        self.grad, self.m, self.v = (torch.zeros_like(self.data) for _ in range(3))
        self.T = self.data[:nT].view(self.rows, self.D)
        self.TL = self.data[nT:nT + nL].view(self.rows, 1)
        self.bias = self.data[nT + nL + pad:nT + nL + pad + 1]
        self.gT = self.grad[:nT].view(self.rows, self.D)
        self.gTL = self.grad[nT:nT + nL].view(self.rows, 1)
        self.gbias = self.grad[nT + nL + pad:nT + nL + pad + 1]
        g = torch.Generator().manual_seed(seed)
        self.T.copy_((torch.randn(self.T.shape, generator=g) * 1e-4).to(self.device))      # nn.init.normal_(std=1e-4), :178
        self.TL.copy_((torch.randn(self.TL.shape, generator=g) * 1e-4).to(self.device))
        dims = [self.F * self.D] + list(hidden_dims)
        blocks = [MLPBlock(i, o, batch_norm, hidden_dropout_rate) for i, o in zip(dims[:-1], dims[1:])]
        blocks.append(nn.Linear(dims[-1], 1))
        self.dnn = nn.Sequential(*blocks).to(self.device)
        for mod in self.dnn.modules():
            if isinstance(mod, nn.Linear):
                nn.init.xavier_normal_(mod.weight)
                nn.init.constant_(mod.bias, 0.0)
        self.mlp_opt = torch.optim.Adam(self.dnn.parameters(), lr=lr, betas=betas, weight_decay=weight_decay)
        self.step = 0
        self.training = True

    # ---- per-field views (reference state-dict granularity)
    def tables(self):
        return [self.T[o:o + c] for o, c in zip(self.offsets.tolist(), self.counts)]

    def tables_lr(self):
        return [self.TL[o:o + c] for o, c in zip(self.offsets.tolist(), self.counts)]

    def train(self, mode=True):
        self.training = mode
        self.dnn.train(mode)
        return self

    def eval(self):
        return self.train(False)

    def encode(self, x):
        """-> (logits [B, 1], E leaf [B, F*D], fm_lr [B]).  DeepFM/main.py:201-209."""
        E, fm_lr = ops.fm_bag_fwd(self.T, self.TL.reshape(-1), self.bias, self.offsets, x)
        E.requires_grad_(self.training)
        logits = fm_lr.unsqueeze(1) + self.dnn(E)
        return logits, E, fm_lr

    def recommend_from_pool(self, x):
        with torch.no_grad():
            return torch.sigmoid(self.encode(x)[0])

    def forward_backward(self, x, labels):
        """loss + every gradient (tables in self.grad, MLP in .grad of its parameters).  DeepFM/main.py:211-215,264-266."""
        logits, E, _ = self.encode(x)
        loss, dlogit, dsum = ops.bce_logits(logits.detach().reshape(-1).contiguous(), labels.reshape(-1).to(torch.float32).contiguous())
        for p in self.dnn.parameters():
            p.grad = None
        logits.backward(dlogit.unsqueeze(1))                 # MLP (aten): parameter grads + dE
        gE, gL = ops.fm_bag_bwd(E.detach(), E.grad.contiguous(), dlogit, self.F, self.D)
        rows = (x + self.offsets.unsqueeze(0)).reshape(-1)
        ops.scatter_add_rows(gE, rows, self.rows, out=self.gT)
        ops.scatter_add_rows(gL, rows, self.rows, out=self.gTL)
        self.gbias.copy_(dsum)
        return loss.squeeze(0)

    def train_step(self, x, labels, max_norm=10.0):
        """forward, backward, clip_grad_norm_(.., 10), Adam with the two weight-decay groups (DeepFM/main.py:187-199,264-268)."""
        loss = self.forward_backward(x, labels)
        mg = [p.grad for p in self.dnn.parameters()]
        total = torch.sqrt(self.grad.pow(2).sum() + sum(g.pow(2).sum() for g in mg))
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
        self.grad.mul_(coef)
        for g in mg:
            g.mul_(coef)
        self.step += 1
        # the LR bias is in the reference's non-embedding group (weight_decay, not embedding_decay): pre-compensate
        self.gbias.add_((self.wd - self.emb_decay) * self.bias)
        ops.adam_step(self.data, self.grad, self.m, self.v, self.step, self.lr, self.betas[0], self.betas[1], 1e-8, self.emb_decay)
        self.mlp_opt.step()
        return loss
