"""Bisect of the wave-per-tile step against the oracle: dropout on/off x long sequences on/off (diagnostic)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sasrec as osas
from recboard_amd.sasrec import SASRecEngine
N, B, S, D, L = 12101, 512, 50, 64, 2
for p, longs in ((0.0, ()), (0.5, ()), (0.0, (17,)), (0.0, (33,)), (0.0, (49,)), (0.5, (49, 33, 17))):
    rng = np.random.default_rng(1)
    w = 1.0 / np.arange(1, N + 1); w /= w.sum()
    lens = np.clip(rng.geometric(1.0 / 5.9, B) + 1, 1, 16)
    for i, v in enumerate(longs): lens[i] = v
    seq, pos, neg = (np.zeros((B, S), np.int64) for _ in range(3))
    for b in range(B):
        n = lens[b]
        seq[b, S - n:] = rng.choice(N, n, p=w) + 1
        pos[b, S - n:] = rng.choice(N, n, p=w)
        neg[b, S - n:] = rng.integers(0, N, n)
    m = SASRecEngine(N, S, D, L, dropout_rate=p, loss="BCE", lr=0.0, weight_decay=0.0, seed=7)
    with torch.no_grad():
        g = torch.Generator().manual_seed(3)
        for k, q in m.params.items():
            if k.endswith("bias"): q.copy_((0.05 * torch.randn(q.shape, generator=g)).cuda())
            elif "LN" in k: q.copy_((1.0 + 0.1 * torch.randn(q.shape, generator=g)).cuda())
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    dev = lambda a: torch.from_numpy(a).cuda()
    seed2 = m._step_seed()
    loss = m.train_step(dev(seq), dev(pos), dev(neg)).item()
    P = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = osas.fit(P, torch.from_numpy(seq), torch.from_numpy(pos), torch.from_numpy(neg), "BCE", L, drop=dict(p=p, seed=seed2))
    ref.backward()
    Gv = m.arena.views(m.arena.grad)
    worst = []
    for k, q in P.items():
        r = q.grad if q.grad is not None else torch.zeros_like(q)
        e = (Gv[k].cpu() - r).abs()
        worst.append((e.max().item() / (r.abs().max().item() + 1e-30), k))
    worst.sort(reverse=True)
    print(f"p={p} longs={longs} loss {loss:.6f} ref {ref.item():.6f}  worst rel errs:", [(f"{a:.1e}", k) for a, k in worst[:4]])
    k = "Item.embeddings.weight"
    e = (Gv[k].cpu() - P[k].grad).abs().max(dim=1).values
    top = torch.topk(e, 5)
    rows = top.indices.tolist()
    print("   worst item rows", rows, [f"{v:.1e}" for v in top.values.tolist()], "row |g|", [f"{P[k].grad[r].abs().max().item():.1e}" for r in rows],
          " in long seqs:", [bool(np.isin(r, seq[:len(longs)])) for r in rows])
