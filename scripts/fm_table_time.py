"""re_fm_table_grad at config 4's shapes, by kind of slice: python scripts/fm_table_time.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from recboard_amd import ops
counts = [94762, 25612, 7, 24, 12, 5, 50, 500, 5000, 50000]
B, D, F, R = 4096, 10, len(counts), sum(counts)
g = torch.Generator(device="cuda").manual_seed(1)
off = torch.tensor([sum(counts[:i]) for i in range(F)], dtype=torch.int64, device="cuda")
x = torch.stack([torch.randint(0, c, (B,), device="cuda", generator=g) for c in counts], 1)
kt = x.t().contiguous().to(torch.int32)
gE, gL = torch.randn(B * F, D, device="cuda"), torch.randn(B * F, 1, device="cuda")
gT, gTL = torch.zeros(R, D, device="cuda"), torch.zeros(R, device="cuda")
sl = ops.fm_table_slices(counts, B, "cuda")
def t(s, name):
    for _ in range(3): ops.fm_table_grad(kt, off, s, gE, gL, gT, gTL)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): ops.fm_table_grad(kt, off, s, gE, gL, gT, gTL)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} slices {s.shape[0]:5d}  {e0.elapsed_time(e1) / 100 * 1e3:7.1f} us (20 launches a graph)")
t(sl, "all")
for f, c in enumerate(counts):
    t(sl[sl[:, 0] == f].contiguous(), f"field {f} ({c} rows)")
t(sl[:1].contiguous(), "one slice of field 0")
