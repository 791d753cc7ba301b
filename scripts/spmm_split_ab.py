"""One LightGCN propagation on the Yelp2018-shaped graph (bench_legs.py): the plain plan against the XCD-split plan and the non-temporal
streams (re_spmm_csr_split), every XCD share 1 .. 7; bit-equality of the results is asserted.   python scripts/spmm_split_ab.py [variant]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench_legs
from recboard_amd import ops
from recboard_amd.gen import LightGCNEngine
from recboard_amd.graph import to_normalized_adj
rng = np.random.default_rng(1)
U, N, eu, ei, wi = bench_legs.yelp_graph(rng)
crow, col, val = to_normalized_adj(U, N, eu, ei)
lg = LightGCNEngine(U, N, crow, col, val, 64, 3)
with torch.no_grad():
    for q in lg.params.values():
        q.normal_(0, 0.1)
nnz = len(col)
ref = torch.empty_like(lg.Xa)
plain = ops.SpmmPlan(lg.crow, 64)
ops.spmm_csr(lg.crow, lg.col, lg.val, plain, lg.X0, ref)
only = sys.argv[1] if len(sys.argv) > 1 else ""
def run(name, plan):
    if only and only != name:
        return
    out = torch.empty_like(ref)
    ops.spmm_csr(lg.crow, lg.col, lg.val, plan, lg.X0, out)
    assert torch.equal(out, ref), name
    t = bench_legs.ev_ms(lambda: ops.spmm_csr(lg.crow, lg.col, lg.val, plan, lg.X0, out), iters=50)
    print(f"{name:28s} {t * 1e3:7.1f} us  {nnz * 256 / t / 1e6:6.0f} GB/s gathered  split {plan.split} share {plan.xcd_share} flags {plan.flags} nlong {plan.nlong}", flush=True)
run("plain", plain)
run("nt", ops.SpmmPlan(lg.crow, 64, 0, True))
for k in range(1, 8):
    for nt in (False, True):
        p = ops.SpmmPlan(lg.crow, 64, U, nt)
        p.xcd_share = k
        run(f"split share {k}{' nt' if nt else ''}", p)
p = ops.SpmmPlan(lg.crow, 64, U)
run("split auto", p)
