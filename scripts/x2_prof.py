"""Kernel-level view of score_topk's split path (for rocprofv3 --kernel-trace --stats): Beauty shape, iid scores, 20 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import lib
if os.environ.get("RECENGINE_LIB"):   # a timing-only ablation build (-DSC_X_...) of the library
    lib.LIB_PATH = os.environ["RECENGINE_LIB"]
from recboard_amd import ops
U, N, D = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (22363, 12101, 64)
g = torch.Generator(device="cuda").manual_seed(1)
q = torch.randn(U, D, device="cuda", generator=g); E = torch.randn(N, D, device="cuda", generator=g)
sp = torch.arange(0, U + 1, device="cuda") * 8
si = torch.sort(torch.randint(0, N, (U, 8), device="cuda", generator=g), 1).values.reshape(-1)
if os.environ.get("X2_STATE"):   # the bench's trained state (scripts/x2_bench_state.py with X2_DUMP=...)
    st = torch.load(os.environ["X2_STATE"])
    q, E, sp, si = (st[k].cuda().contiguous() for k in ("q", "E", "sp", "si"))
for _ in range(20):
    ops.score_topk(q, E, sp, si, 50)
torch.cuda.synchronize()
