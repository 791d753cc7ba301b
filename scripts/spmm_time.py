"""One LightGCN propagation (re_spmm_csr) on the Yelp2018-shaped graph of bench_legs.py: launch time and bandwidths."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_legs
from recboard_amd.gen import LightGCNEngine
from recboard_amd.graph import to_normalized_adj
rng = np.random.default_rng(1)
U, N, eu, ei, wi = bench_legs.yelp_graph(rng)
crow, col, val = to_normalized_adj(U, N, eu, ei)
lg = LightGCNEngine(U, N, crow, col, val, 64, 3)
with torch.no_grad():
    for q in lg.params.values():
        q.normal_(0, 0.1)
t = bench_legs.ev_ms(lambda: lg._spmm(lg.X0, lg.Xa), iters=50)
nnz = len(col)
print(f"spmm {t * 1e3:.1f} us  {nnz * 256 / t / 1e6:.0f} GB/s of gathered X rows  {(nnz * 12 + (U + N) * 520) / t / 1e6:.0f} GB/s HBM stream", flush=True)
