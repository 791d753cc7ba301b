"""Times re_scatter_add_rows on the training-batch shapes (SASRec/Beauty: 76 800 x 64 -> 12 102 rows; DeepFM-like scalar)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from recboard_amd import ops

def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

rng = np.random.default_rng(0)
for (n, D, R, zipf) in [(76800, 64, 12102, 1.2), (76800, 64, 12102, 0), (160000, 10, 1000000, 1.1), (160000, 1, 1000000, 1.1), (1536, 64, 50000, 0), (400000, 64, 12102, 1.2)]:
    if zipf:
        idx = np.minimum(rng.zipf(zipf, n), R - 1)
    else:
        idx = rng.integers(0, R, n)
    idx = torch.from_numpy(idx.astype(np.int64)).cuda()
    g = torch.randn(n, D, device="cuda")
    out = torch.empty(R, D, device="cuda")
    ws = torch.empty(ops.lib.load().re_scatter_add_rows_workspace_bytes(n, D, R), dtype=torch.uint8, device="cuda")
    f = lambda: ops.scatter_add_rows(g, idx, R, 0, 1.0, out=out, ws=ws)
    us = timeit(f)
    ref = torch.zeros(R, D, device="cuda", dtype=torch.float64).index_add_(0, idx, g.double())
    ref[0] = 0
    err = (out.double() - ref).abs().max().item()
    o2 = out.clone(); f()
    print(f"n={n} D={D} R={R} zipf={zipf}: {us:7.1f} us   max|err| vs f64 {err:.2e}  reproducible={torch.equal(o2, out)}")
