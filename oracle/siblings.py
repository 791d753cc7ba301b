"""Oracle: DCN and SimGCL forward (TEST INFRASTRUCTURE: only tests/ may import this).  torch-CPU fp32 restatements, pinned against
tests/golden/dcn.npz and simgcl.npz (made by importing the reference's own model files: tests/golden/make_golden.py).

  DCN     DCN/main.py:153-165   emb = cat_f E_f[x_f]; deep = MLP(emb) (Linear -> BatchNorm1d -> ReLU -> Dropout per block, :48-69);
                                cross_0 = emb, cross_{i+1} = (cross_i w_i) * emb + b_i (:34-46); logit = fc(cat(deep, cross))
  SimGCL  SimGCL/main.py:98-147 avg = sum_l (Adj^l X0) / L (no layer-0 term); rec = BPR(<u, i+>, <u, i->) on avg rows; emb =
                                (|U0[u]|^2 + |I0[i+]|^2 + |I0[i-]|^2) / 2 / B; ssl = CE(norm(u)[users] norm(u)[users]^T / tau, arange)
                                + the same for items[positives]  (eps = 0: the two views coincide)
  JGCF    JGCF/main.py:103-134, modules.py:8-83  Jacobi recurrence z_l on the dense adjacency; low = mean_l(coef_l z_l); tables [low | w X - low]
  BERT4Rec BERT4Rec/main.py:166-186  x = LN(E[seq] + P); per block (post-norm nn.TransformerEncoderLayer, GELU): x = LN1(x + MHA(x; pad keys
                                masked)), x = LN2(x + W2 gelu(W1 x)); logits = fc(x)[masked positions]; CE(logits, the items that were masked)
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


def bert4rec_states(sd, seq, num_blocks, num_heads, eps=1e-5):
    """sd: the reference's state_dict (tensors); seq [B, S] (0 = pad, 1 = mask token).  -> encoder states [B, S, D]."""
    E, P = sd["Item.embeddings.weight"], sd["Position.weight"]
    B, S = seq.shape
    D = E.shape[1]
    dh = D // num_heads

    def ln(x, pre):
        mu = x.mean(-1, keepdim=True)
        var = ((x - mu) ** 2).mean(-1, keepdim=True)
        return (x - mu) / torch.sqrt(var + eps) * sd[pre + ".weight"] + sd[pre + ".bias"]

    x = ln(E[seq] + P[:S].unsqueeze(0), "layernorm")
    pad = seq == 0
    for l in range(num_blocks):
        pre = f"encoder.layers.{l}."
        qkv = x @ sd[pre + "self_attn.in_proj_weight"].t() + sd[pre + "self_attn.in_proj_bias"]
        q, k, v = (t.reshape(B, S, num_heads, dh).transpose(1, 2) for t in qkv.split(D, dim=-1))
        sc = (q @ k.transpose(-1, -2)) / dh ** 0.5
        sc = sc.masked_fill(pad[:, None, None, :], float("-inf"))
        a = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, S, D)
        a = a @ sd[pre + "self_attn.out_proj.weight"].t() + sd[pre + "self_attn.out_proj.bias"]
        x = ln(x + a, pre + "norm1")
        f = F.gelu(x @ sd[pre + "linear1.weight"].t() + sd[pre + "linear1.bias"]) @ sd[pre + "linear2.weight"].t() + sd[pre + "linear2.bias"]
        x = ln(x + f, pre + "norm2")
    return x


def bert4rec_loss(sd, seq, rnds, mask_ratio, num_blocks, num_heads):
    masked = torch.where(rnds < mask_ratio, torch.ones_like(seq), seq).masked_fill(seq == 0, 0)
    m = masked == 1
    h = bert4rec_states(sd, masked, num_blocks, num_heads)
    return F.cross_entropy(h[m] @ sd["fc.weight"].t() + sd["fc.bias"], seq[m])


def jgcf_tables(U0, I0, crow, col, val, L, alpha, beta, scaling, weight4mid, gammas):
    n = U0.shape[0] + I0.shape[0]
    A = torch.sparse_csr_tensor(crow, col, val, size=(n, n)).to_dense()
    x = torch.cat((U0, I0), 0)
    zs = [x]
    for l in range(1, L + 1):
        if l == 1:
            zs.append((alpha - beta) / 2 * x + (alpha + beta + 2) / 2 * (A @ x))
            continue
        s = 2 * l + alpha + beta
        c0, c1 = 2 * l * (l + alpha + beta) * (s - 2), (s - 1) * (alpha ** 2 - beta ** 2)
        c2, c3 = (s - 1) * s * (s - 2), 2 * (l + alpha - 1) * (l + beta - 1) * s
        zs.append((c1 * zs[-1] + c2 * (A @ zs[-1]) - c3 * zs[-2]) / c0)
    coefs = (torch.tanh(gammas) * scaling).cumprod(0)            # [L + 1, 1]
    low = (torch.stack(zs, 1) * coefs).mean(1)
    t = torch.cat((low, weight4mid * x - low), 1)
    return t[: U0.shape[0]], t[U0.shape[0]:]
