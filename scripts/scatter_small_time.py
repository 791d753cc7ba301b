"""re_scatter_add_rows_small by key distribution (run under scripts/prof_any.sh: kernel durations from the trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from recboard_amd import ops
R, D, NR = 12102, 64, 32768
g = torch.Generator(device="cuda").manual_seed(1)
G = torch.randn(3, NR, D, device="cuda", generator=g)
out = torch.empty(R, D, device="cuda")
w = 1.0 / np.arange(1, R); w /= w.sum()
rng = np.random.default_rng(0)
for name, n, mk in (("uniform", 4496, lambda n: rng.integers(1, R, (3, n))),
                    ("zipf", 4496, lambda n: rng.choice(R - 1, (3, n), p=w) + 1),
                    ("onerow", 4496, lambda n: np.full((3, n), 7)),
                    ("empty", 4496, lambda n: np.zeros((3, n), np.int64)),
                    ("uniform_full", 25600, lambda n: rng.integers(1, R, (3, n)))):
    keys = torch.zeros(3, NR, dtype=torch.int32, device="cuda")
    keys[:, :n] = torch.from_numpy(mk(n).astype(np.int32)).cuda()
    for _ in range(5):
        ops.scatter_add_rows_small(G, keys, R, out, n_regions=3, region_stride=NR, n=n)
    torch.cuda.synchronize()
    print(name)
