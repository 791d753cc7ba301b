"""Oracle: DCN and SimGCL forward (TEST INFRASTRUCTURE: only tests/ may import this).  torch-CPU fp32 restatements, pinned against
tests/golden/dcn.npz and simgcl.npz (made by importing the reference's own model files: tests/golden/make_golden.py).

  DCN     DCN/main.py:153-165   emb = cat_f E_f[x_f]; deep = MLP(emb) (Linear -> BatchNorm1d -> ReLU -> Dropout per block, :48-69);
                                cross_0 = emb, cross_{i+1} = (cross_i w_i) * emb + b_i (:34-46); logit = fc(cat(deep, cross))
  SimGCL  SimGCL/main.py:98-147 avg = sum_l (Adj^l X0) / L (no layer-0 term); rec = BPR(<u, i+>, <u, i->) on avg rows; emb =
                                (|U0[u]|^2 + |I0[i+]|^2 + |I0[i-]|^2) / 2 / B; ssl = CE(norm(u)[users] norm(u)[users]^T / tau, arange)
                                + the same for items[positives]  (eps = 0: the two views coincide)
"""
import torch
import torch.nn.functional as F


def dcn_logits(tables, x, dnn, cross, fc, eps=1e-5):
    """tables: list of [count_f, D]; x [B, F] int64; dnn: list of (W, b, gamma, beta) per block (train-mode BatchNorm: batch statistics);
    cross: list of (w [1, Din], bias [Din]); fc: (W [1, Din + H], b [1]).  -> logits [B, 1]."""
    emb = torch.cat([t[x[:, f]] for f, t in enumerate(tables)], dim=1)
    h = emb
    for W, b, gamma, beta in dnn:
        z = h @ W.t() + b
        mean, var = z.mean(0), z.var(0, unbiased=False)
        h = torch.relu((z - mean) / torch.sqrt(var + eps) * gamma + beta)
    c = emb
    for w, bias in cross:
        c = (c @ w.t()) * emb + bias
    return torch.cat((h, c), dim=1) @ fc[0].t() + fc[1]


def simgcl_losses(U0, I0, crow, col, val, users, pos, neg, num_layers, temperature):
    n = U0.shape[0] + I0.shape[0]
    A = torch.sparse_csr_tensor(crow, col, val, size=(n, n)).to_dense()
    x = torch.cat((U0, I0), 0)
    avg = torch.zeros_like(x)
    for _ in range(num_layers):
        x = A @ x
        avg = avg + x / num_layers
    ue, ie = avg[: U0.shape[0]], avg[U0.shape[0]:]
    u, ip, ineg = ue[users], ie[pos], ie[neg]
    rec = F.softplus((u * ineg).sum(-1) - (u * ip).sum(-1)).mean()
    emb = (U0[users].pow(2).sum() + I0[pos].pow(2).sum() + I0[neg].pow(2).sum()) / 2 / users.numel()
    t = torch.arange(users.numel())
    un, inn = F.normalize(ue, dim=-1)[users], F.normalize(ie, dim=-1)[pos]
    ssl = F.cross_entropy(un @ un.t() / temperature, t) + F.cross_entropy(inn @ inn.t() / temperature, t)
    return rec, emb, ssl, ue, ie
