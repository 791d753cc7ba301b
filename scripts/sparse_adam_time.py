"""re_sparse_adam_rows (sort + segmented sum) vs re_sparse_adam_rows_small (owner-computes, one launch) on a config-5-shaped step:
3 regions of compact rows, Zipf(1.05) sequence / positive items, uniform negatives, a table far larger than the caches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recboard_amd import ops

R, D = int(os.environ.get("ROWS", 20_000_000)), 128
NR, live = 32768, 16 * 310
rng = np.random.default_rng(1)
W = torch.randn(R, D, device="cuda"); m = torch.zeros_like(W); v = torch.zeros_like(W)
g = torch.randn(3 * NR, D, device="cuda")
n_dev = torch.tensor([310], dtype=torch.int32, device="cuda")


def keys_for(dist):
    k = np.zeros((3, NR), np.int32)
    if dist == "zipf":
        k[0, :live] = np.minimum(rng.zipf(1.05, live), R - 1)
        k[1, :live] = np.minimum(rng.zipf(1.05, live), R - 1)
    else:
        k[0, :live] = rng.integers(1, R, live)
        k[1, :live] = rng.integers(1, R, live)
    k[2, :live] = rng.integers(1, R, live)
    k[:, :live][rng.random((3, live)) < 0.3] = 0          # slot padding inside the tiles
    return torch.from_numpy(k).cuda()


def timed(f, reps=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for dist in ("uniform", "zipf"):
    keys = keys_for(dist)
    k64 = keys.view(-1).long()
    ws = torch.empty(ops.lib.load().re_scatter_add_rows_workspace_bytes(k64.numel(), D, R), dtype=torch.uint8, device="cuda")
    t_old = timed(lambda: ops.sparse_adam_rows(g, k64, W, m, v, 3, 1e-3, padding_idx=0, ws=ws))
    t_new = timed(lambda: ops.sparse_adam_rows_small(g, keys, W, m, v, step=3, lr=1e-3, padding_idx=0, n_dev=n_dev, n_mul=16))
    t_new_all = timed(lambda: ops.sparse_adam_rows_small(g, keys, W, m, v, step=3, lr=1e-3, padding_idx=0))
    u, c = torch.unique(k64[k64 > 0], return_counts=True)
    print(f"{dist}: sorted {t_old:.1f} us   small (live rows) {t_new:.1f} us   small (all {3 * NR} keys) {t_new_all:.1f} us   distinct {u.numel()} max run {int(c.max())}",
          flush=True)
