"""Time of re_sparse_adam_rows_small on config-5-shaped key lists: Zipf(1.05) items in two of the three regions vs all-uniform keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from recboard_amd import ops
R, D, NR, live = int(os.environ.get("SA_ROWS", "20000001")), 128, 4096, 3616
NSETS = int(os.environ.get("SA_SETS", "8"))   # distinct key sets cycled through (8: the rows stay in the Infinity Cache; 64+: cold)
rng = np.random.default_rng(3)
W = torch.zeros((R, D), device="cuda"); m = torch.zeros_like(W); v = torch.zeros_like(W)
g = torch.randn((3 * NR, D), device="cuda")
hyper = torch.tensor([1e-3, 1.0], device="cuda")
n_dev = torch.tensor([live // 16], dtype=torch.int32, device="cuda")
w = 1.0 / np.arange(1, 2_000_001) ** 1.05; w /= w.sum()          # (Zipf over the first 2 M ids: the head is what matters here)
cdf = np.cumsum(w)
def zipf(n):
    return (np.searchsorted(cdf, rng.random(n)) + 1).astype(np.int32)
for name in ("zipf", "uniform", "zipf_top_removed"):
    ks = []
    for rep in range(NSETS):
        k = np.zeros((3, NR), np.int32)
        for r in range(3):
            k[r, :live] = rng.integers(1, R, live) if (name == "uniform" or r == 2) else zipf(live)
        if name == "zipf_top_removed":
            k[(k > 0) & (k <= 4)] = 17
        ks.append(torch.from_numpy(k).cuda())
    i = [0]
    def f():
        ops.sparse_adam_rows_small(g, ks[i[0] % NSETS], W, m, v, hyper=hyper, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=1e-6, padding_idx=0, n_dev=n_dev, n_mul=16)
        i[0] += 1
    print(name, "%.1f us" % (1e3 * bench.graph_time_ms(f, reps=NSETS, iters=3)), "| contributions of the most frequent key:", int(np.bincount(ks[0].cpu().numpy().ravel())[1:].max()))
